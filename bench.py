#!/usr/bin/env python
"""
bench.py — throughput of the RDF hot path (atom-pairs/s) on N GPUs of one node.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step is one pass of `_rdf_loop` (structural/rdf_cn.py:72-97 of the reference) over one batch of
synthetic frames that are already resident in HBM: BASELINE.json configs[1] — 10 000 atoms x 200
frames, cubic box L = 50 A, 4 atom types, all 10 type pairs, r_cut 20 A, 400 bins. For N > 1 every
rank owns its own 200 frames (weak scaling) and the frame-summed uint64 histograms are
all-reduced over RCCL in every step, inside the timed region.

Rank 0 prints ONE JSON line. Extra keys: `roofline` (dominant kernel, measured with HIP events on the
launch stream inside the library), `cpu_baseline` (oracle/cpu_ref.c on the host cores, bounded
sample), `msd` (frame-pairs/s of the single-origin MSD kernel, HBM-bound, reported beside the RDF
number because BASELINE.json's metric names both).
"""

import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

FP64_NONFUSED_PEAK = 39.3e12  # 256 CU x 128 lanes x 2.4 GHz / 2 (FMA is forbidden by parity)
HBM_PEAK = 8.0e12             # spec, /opt/skills/guides/MI355X_MICROARCH.md
OPS_PER_PAIR = 18             # SURVEY.md §8d algorithmic FP64 ops per atom pair


def cpu_baseline(cfg, types, rel, n_sample_frames):
    """oracle/cpu_ref.c (single thread, -O2 -ffp-contract=off) on the same workload, bounded sample."""
    from oracle import cref
    from mdproptools_amd import synth

    cref.build()
    n = cfg["n_atoms"]
    xyz = synth.rdf_frames(n, range(n_sample_frames), cfg["box_len"], cfg["seed_offset"])
    L = [cfg["box_len"]] * 3
    t0 = time.perf_counter()
    for f in range(n_sample_frames):
        cref.rdf_pairs(xyz[f], types, rel, L, cfg["r_cut"] ** 2, cfg["bin_size"], 400)
    dt = time.perf_counter() - t0
    pairs = n_sample_frames * n * (n - 1) // 2
    return {
        "value": pairs / dt, "unit": "atom-pairs/s", "cores": 1, "kind": "port",
        "sample": "%d of %d frames of the same workload (cost is linear in frames), %.1f s, "
                  "oracle/cpu_ref.c gcc -O2 single thread; host has %d cores"
                  % (n_sample_frames, cfg["n_frames"], dt, os.cpu_count() or 0),
    }


def cpu_baseline_all_cores(cfg, types, rel):
    """The same loop frame-parallel over every host core (frames are independent; the reference itself only does
    this in get_charge_flux, conductivity.py:190): one frame per core on a thread pool — the ctypes call into
    oracle/cpu_ref.c releases the GIL — wall time of the whole pool."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import cref
    from mdproptools_amd import synth

    cores = min(os.cpu_count() or 1, 64)  # bounded sample: 64 threads, one frame each
    n, L = cfg["n_atoms"], cfg["box_len"]
    frames = synth.rdf_frames(n, range(cores), L, cfg["seed_offset"])

    def one(f):
        cref.rdf_pairs(frames[f], types, rel, [L] * 3, cfg["r_cut"] ** 2, cfg["bin_size"], 400)

    with ThreadPoolExecutor(max_workers=cores) as pool:
        t0 = time.perf_counter()
        list(pool.map(one, range(cores)))
        wall = time.perf_counter() - t0
    return {"value": cores * (n * (n - 1) // 2) / wall, "unit": "atom-pairs/s", "cores": cores, "kind": "port",
            "sample": "%d frames on %d threads (one each, around the C oracle; host reports %d cores), %.1f s wall"
                      % (cores, cores, os.cpu_count() or 0, wall)}


def msd_leg(B, torch, device, steps):
    """Single-origin MSD (diffusion.py:212-218) on a resident random walk: frame-pairs/s and HBM GB/s."""
    from mdproptools_amd import synth

    E, F = 50_000, 256
    r = torch.from_numpy(synth.random_walk(E, F)).to(device)
    pairs = [(0, t) for t in range(F)]
    B.msd_pairs(r, pairs, [0, E], scale=1e-10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kms = 0.0
    for _ in range(steps):
        B.msd_pairs(r, pairs, [0, E], scale=1e-10)
        kms += B.default_context().last_kernel_ms()[0]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    alg_bytes = 24.0 * E * F  # 24*E bytes per frame pair (origin frame amortised)
    return {
        "metric": "frame-pairs/s", "value": steps * F / dt, "workload": "50k entities x 256 frames, pairs (0,t)",
        "kernel_ms": kms / steps,
        "roofline": {"bound": "hbm", "achieved": alg_bytes / (kms / steps * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9,
                     "unit": "GB/s", "frac": alg_bytes / (kms / steps * 1e-3) / HBM_PEAK, "traffic": None},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=10)
    ap.add_argument("--variant", type=int, default=None, help="kernel variant knob (A/B only)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0 and world == 1 and args.gpus > 1:
            print("bench.py: --gpus %d needs torch.distributed.run with %d ranks" % (args.gpus, args.gpus),
                  file=sys.stderr)
            sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the hot path has no CPU fallback", file=sys.stderr)
        sys.exit(1)
    # one process per GPU; MDHIP_DIST_BACKEND=gloo lets several ranks share one GPU (only used to exercise
    # the N > 1 code path on a 1-GPU box; the driver's scaling runs use the default, RCCL)
    backend = os.environ.get("MDHIP_DIST_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from mdproptools_amd import backend as B
    from mdproptools_amd import synth
    from mdproptools_amd._lib import default_context

    ctx = default_context(dev_index)
    if args.variant is not None:
        ctx.set_option("rdf_variant", args.variant)

    cfg = synth.rdf_config("C2")
    n, F, L = cfg["n_atoms"], cfg["n_frames"], cfg["box_len"]
    nb = int(cfg["r_cut"] / cfg["bin_size"])
    types = synth.rdf_types(n)
    rel = np.array(synth.ALL_PAIRS_4, dtype=np.int32)
    box = np.full((F, 3), L)
    frame_ids = range(rank * F, (rank + 1) * F)  # every rank its own frames: weak scaling
    xyz = torch.from_numpy(synth.rdf_frames(n, frame_ids, L, cfg["seed_offset"])).to(device)
    pairs_per_step = F * n * (n - 1) // 2

    from mdproptools_amd import dist as D

    def local_pass(x, t, b, rl, rc, dd, nbins):
        return B.rdf_loop(x, t, b, rl, rc, dd, nbins, per_frame=False, ctx=ctx)

    def step():
        # N > 1: frame shards per rank, one RCCL all-reduce of the uint64 histograms per step
        # (mdproptools_amd/dist.py). The collective of step k is left in flight while step k+1 computes and is
        # waited for right after: every step's sums are complete inside the timed region.
        return D.rdf_sharded_async(xyz, types, box, rel, cfg["r_cut"], cfg["bin_size"], nb, compute=local_pass)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step().wait()
    fence()
    kernel_ms, aux_ms, launches = 0.0, 0.0, 0
    pending = None
    t0 = time.perf_counter()
    for _ in range(args.steps):
        handle = step()
        ms, nl = ctx.last_kernel_ms()
        kernel_ms += ms
        aux_ms += ctx.last_aux_ms()
        launches += nl
        if pending is not None:
            pending.wait()
        pending = handle
    full, part, _ov = pending.wait()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # sanity inside the bench: the result of the last step is a real histogram of the right size
    expect_in = 4.0 / 3.0 * np.pi * cfg["r_cut"] ** 3 / L ** 3
    frac_in = float(full.sum()) / 2.0 / (world * pairs_per_step)
    assert abs(frac_in - expect_in) < 0.01 * expect_in, (frac_in, expect_in)

    if rank == 0:
        value = world * pairs_per_step * args.steps / elapsed
        kdur = kernel_ms / max(launches, 1) * 1e-3  # average duration of one pair_hist launch
        alg_ops = pairs_per_step * OPS_PER_PAIR
        out = {
            "metric": "atom-pairs/s", "value": value, "unit": "atom-pairs/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32/f64",
            "data": "synthetic",
            "config": {"workload": "C2: 10k atoms x 200 frames per GPU, cubic L=50 A, 4 types, 10 type-pair "
                                   "relations, r_cut 20 A, 400 bins, frame-summed uint64 histograms"
                                   + (", RCCL all-reduce per step" if world > 1 else ""),
                       "pairs_per_step_per_gpu": pairs_per_step, "kernel_variant": ctx_variant(ctx, args)},
            "roofline": {
                "bound": "fp64-valu",
                "note": "neither hbm nor mfma bounds this kernel: 28 B and 18 unfused FP64 ops per atom pair the "
                        "reference evaluates (SURVEY.md 8d); peak = 256 CU x 128 lanes x 2.4 GHz / 2; the hbm view is "
                        "given beside it. achieved counts ALGORITHMIC ops: the sweep culls ~15 % of the pairs "
                        "spatially and classifies the rest with packed f32 arithmetic (two pairs per VALU slot; the "
                        "~0.2 % of pairs inside the error band of a bin edge are resolved by the exact f64 chain, so "
                        "the integers are the reference's), hence frac > 1 against the unfused-FP64 roof. What binds "
                        "is VALU issue: see valu_issue (instruction count from profiles/r01_pmc_summary.txt)",
                "kernel": ctx.last_kernel_name(),
                "launch_ms": kdur * 1e3, "prepass_ms_per_step": aux_ms / args.steps,
                "achieved": alg_ops / kdur / 1e12, "peak": FP64_NONFUSED_PEAK / 1e12, "unit": "TFLOP/s",
                "frac": alg_ops / kdur / FP64_NONFUSED_PEAK,
                "hbm": {"achieved": 28.0 * n * F / kdur / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                        "frac": 28.0 * n * F / kdur / HBM_PEAK},
                "traffic": load_traffic(),
                "valu_issue": valu_issue(kdur),
            },
        }
        try:
            out["msd"] = msd_leg(B, torch, device, max(3, args.steps // 4))
        except Exception as e:  # the MSD leg is informative; the RDF line must still be printed
            out["msd"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, types, rel, args.cpu_frames)
            try:
                out["cpu_baseline"]["all_cores"] = cpu_baseline_all_cores(cfg, types, rel)
            except Exception as e:  # informative only
                out["cpu_baseline"]["all_cores"] = {"error": repr(e)}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def ctx_variant(ctx, args):
    """Kernel variant in use: the library default is 1 (fast kernel); --variant overrides it for A/B runs."""
    return int(os.environ.get("MDHIP_RDF_VARIANT", "1")) if args.variant is None else int(args.variant)


def valu_issue(kdur):
    """VALU issue-slot view of the pair kernel: wave-instructions per launch (a constant of this workload, from the
    committed rocprofv3 --pmc run, profiles/rdf_traffic.json) over the live launch duration, against one VALU
    instruction per SIMD every 4 cycles (256 CU x 4 SIMD x 2.4 GHz / 4)."""
    p = os.path.join(HERE, "profiles", "rdf_traffic.json")
    try:
        insts = float(json.load(open(p))["valu_wave_instructions_per_launch"])
    except Exception:
        return None
    peak = 256 * 4 * 2.4e9 / 4
    return {"achieved": insts / kdur / 1e9, "peak": peak / 1e9, "unit": "G wave-instructions/s",
            "frac": insts / kdur / peak}


def load_traffic():
    """HBM bytes per launch from a committed rocprofv3 --pmc run of this same command (profiles/), or None."""
    p = os.path.join(HERE, "profiles", "rdf_traffic.json")
    if os.path.exists(p):
        try:
            return json.load(open(p)).get("hbm_bytes_per_launch")
        except Exception:
            return None
    return None


if __name__ == "__main__":
    main()
