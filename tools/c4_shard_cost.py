#!/usr/bin/env python
"""tools/c4_shard_cost.py [world] — what ONE rank of `world` does per fused C4 step (dist.msd_step_sharded_async), on one
GPU: the shard rank 0 would hold (F / world frames x E entities, F frames x E / world entities), the same library calls,
tensor operations and copies, and the two collectives through a ONE-rank RCCL group (their launch cost, not their wire
time). Steps pipelined as bench.py --workload c4 does. Prints step time against the kernels' time: the host-bound floor
of the strong-scaled step."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from mdproptools_amd import backend as B  # noqa: E402
from mdproptools_amd import dist as D  # noqa: E402
from mdproptools_amd._lib import default_context  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
E, F, tao = int(os.environ.get("C4_E", "50000")), int(os.environ.get("C4_F", "5000")), 4  # (tiny sizes: the host floor)
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
ctx = default_context(0)
lo, hi = D.frame_shard(F, 0, world)
e_lo, e_hi = D.entity_shard(E, 0, world)
g = torch.Generator(device=dev)
g.manual_seed(5)
r_f = torch.cumsum(torch.randn((hi - lo, 3, E), generator=g, device=dev, dtype=torch.float64) * 0.1, dim=0)
r_e = torch.cumsum(torch.randn((F, 3, e_hi - e_lo), generator=g, device=dev, dtype=torch.float64) * 0.1, dim=0)
goff = np.array([0, E], dtype=np.int64)
G, n_lags = 1, F
nS, nW, nL = F * G * 4, E * 4, n_lags * G * 4
S = D._step_stream(dev, ctx)
S2 = D._step_stream(dev, ctx, post=True)
w = torch.from_numpy((F - np.arange(n_lags)).astype(np.float64)[:, None] * float(e_hi - e_lo)).to(dev)
kern = []


def issue_two_waits():
    with torch.cuda.stream(S):
        zero = torch.zeros((3, E), dtype=torch.float64, device=dev)
        mine = torch.stack([r_f[0], r_f[hi - lo - 1]])
        allf = torch.empty((1,) + tuple(mine.shape), dtype=mine.dtype, device=dev)
        dist.all_gather_into_tensor(allf, mine)
        r0 = allf[0, 0].contiguous()
        res = torch.zeros(nS + nW + nL, dtype=torch.float64, device=dev)
        single = res[:nS].view(F, G, 4)
        win = res[nS:nS + nW].view(E, 4)
        means = torch.empty((n_lags, 1, 4), dtype=torch.float64, device=dev)
        hs = [B.msd_origin(r_f, r0, goff, scale=1e-10, out=single[lo:hi], ctx=ctx, async_=True),
              B.msd_windows(r_f, tao, scale=1e-10, out=win, ctx=ctx, async_=True),
              B.lag_msd(r_e, F - 1, [0, e_hi - e_lo], scale=1.0, out=means, ctx=ctx, async_=True)]
    return hs, res, means, (zero, allf, r0)


def collect_two_waits(st):
    hs, res, means, _keep = st
    with torch.cuda.stream(S2):
        k = 0.0
        for h in hs:
            h.wait()
            k += sum(h.stats()[:2])
        kern.append(k)
        res[nS + nW:].view(n_lags, G, 4).copy_(means * w[:, :, None])
        dist.all_reduce(res)
        return res.cpu().numpy()



def issue():
    # the order of dist.msd_step_sharded_async (round 5): the three calls, what follows them on the device, the all-reduce
    # and the copy to page-locked host memory are all QUEUED here; collect() waits once
    with torch.cuda.stream(S):
        zero = torch.zeros((3, E), dtype=torch.float64, device=dev)
        mine = torch.stack([r_f[0], r_f[hi - lo - 1]])
        allf = torch.empty((1,) + tuple(mine.shape), dtype=mine.dtype, device=dev)
        dist.all_gather_into_tensor(allf, mine)
        r0 = allf[0, 0].contiguous()
        res = torch.zeros(nS + nW + nL + 2, dtype=torch.float64, device=dev)
        single = res[:nS].view(F, G, 4)
        win = res[nS:nS + nW].view(E, 4)
        means = torch.empty((n_lags, 1, 4), dtype=torch.float64, device=dev)
        ctx.set_option("lag_variant", 2)
        hs = [B.msd_origin(r_f, r0, goff, scale=1e-10, out=single[lo:hi], ctx=ctx, async_=True),
              B.msd_windows(r_f, tao, scale=1e-10, out=win, ctx=ctx, async_=True),
              B.lag_msd(r_e, F - 1, [0, e_hi - e_lo], scale=1.0, out=means, ctx=ctx, async_=True,
                        status_out=res[nS + nW + nL + 1:])]
        ctx.set_option("lag_variant", -1)
        ev = torch.cuda.Event()
        ev.record(S)
    S2.wait_event(ev)
    with torch.cuda.stream(S2):
        res[nS + nW:nS + nW + nL].view(n_lags, G, 4).copy_(means * w[:, :, None])
        dist.all_reduce(res, group=D._post_group())
        flat_np, flat_t = D._pinned_like(res)
        flat_t.copy_(res, non_blocking=True)
        done = torch.cuda.Event()
        done.record(S2)
    return hs, res, means, (zero, allf, r0), done, flat_np


def collect(st):
    hs, res, means, _keep, done, flat_np = st
    done.synchronize()
    k = 0.0
    for h in hs:
        h.wait()
        k += sum(h.stats()[:2])
    kern.append(k)
    assert flat_np[nS + nW + nL + 1] <= 1e-10
    return flat_np



if os.environ.get("C4_ORDER", "two") == "two":  # (the default order of dist.msd_step_sharded_async; C4_ORDER=one: the opt-in one)
    issue, collect = issue_two_waits, collect_two_waits

t_issue, t_collect = [], []
_issue, _collect = issue, collect


def issue():  # noqa: F811
    t = time.perf_counter()
    st = _issue()
    t_issue.append(time.perf_counter() - t)
    return st


def collect(st):  # noqa: F811
    t = time.perf_counter()
    out = _collect(st)
    t_collect.append(time.perf_counter() - t)
    return out


prev = None
for _ in range(5):
    st = issue()
    if prev is not None:
        collect(prev)
    prev = st
collect(prev)
torch.cuda.synchronize()
kern.clear()
t_issue.clear()
t_collect.clear()
t0 = time.perf_counter()
prev = None
for _ in range(steps):
    st = issue()
    if prev is not None:
        collect(prev)
    prev = st
collect(prev)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps * 1e3
print("world %d shard (frames %d, entities %d): %.3f ms per step, kernels %.3f ms, step / kernels %.2f" % (
    world, hi - lo, e_hi - e_lo, dt, float(np.mean(kern)), dt / float(np.mean(kern))), flush=True)
print("   host time inside issue() %.3f ms, inside collect() %.3f ms (collect includes waiting for the step's kernels)" % (
    float(np.median(t_issue)) * 1e3, float(np.median(t_collect)) * 1e3), flush=True)
dist.destroy_process_group()
