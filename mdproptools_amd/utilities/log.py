"""Concatenate the thermo tables of several LAMMPS logs — /root/reference/mdproptools/utilities/log.py:10-28."""

import glob
import os
import re

import pandas as pd

from ..io import parse_lammps_log


def concat_log(log_pattern, step=None, working_dir=None):
    working_dir = working_dir or os.getcwd()
    files = glob.glob(f"{working_dir}/{log_pattern}")
    if len(files) > 1:
        rx = re.compile(".*" + log_pattern.replace("*", "([0-9]+)").replace("\\", "\\\\"))
        files = sorted(files, key=lambda f: int(rx.match(f).group(1)))
    logs = [parse_lammps_log(f)[0] for f in files]
    # the last row of a log is the first row of the next one
    logs = [lg[:-1] for lg in logs[:-1]] + logs[-1:]
    full_log = pd.concat(logs, ignore_index=True)
    if step:
        full_log = full_log.loc[range(1, full_log.shape[0], 50000)]  # log.py:25-27: fixed stride
    return full_log
