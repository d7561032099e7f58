"""
Thermo tables of several LAMMPS log files as one DataFrame.

Role of the reference's `concat_log` (/root/reference/mdproptools/utilities/log.py:10-28), which
`Diffusion.get_msd_from_log` uses: restart segments `log.1`, `log.2`, ... each repeat the previous
segment's final thermo row as their first one, so every segment but the last gives up its final row
before the tables are stacked.
"""

import glob
import os
import re

import pandas as pd

from ..io import parse_lammps_log


def _segment_number(path, log_pattern):
    """The integer a '*' in `log_pattern` stands for in `path` (restart segments sort numerically)."""
    rx = ".*" + log_pattern.replace("*", "([0-9]+)").replace("\\", "\\\\")
    return int(re.match(rx, path).group(1))


def concat_log(log_pattern, step=None, working_dir=None):
    """
    log_pattern: file name or '*' pattern below working_dir (default: cwd). Returns the first run's
    thermo table of every matching file, stacked in segment order with a fresh index. A truthy `step`
    thins the result to rows 1, 50001, 100001, ... (the reference's fixed stride).
    """
    folder = os.getcwd() if working_dir is None else working_dir
    paths = glob.glob(f"{folder}/{log_pattern}")
    if len(paths) > 1:
        paths.sort(key=lambda p: _segment_number(p, log_pattern))
    tables = []
    for k, path in enumerate(paths):
        table = parse_lammps_log(path)[0]
        is_last = k == len(paths) - 1
        tables.append(table if is_last else table.iloc[:-1])
    merged = pd.concat(tables, ignore_index=True)
    if step:
        merged = merged.loc[range(1, merged.shape[0], 50000)]
    return merged
