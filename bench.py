#!/usr/bin/env python
"""Headline benchmark: PSMF filter timesteps/s at d=100k, r=32 (BASELINE.json `metric`).

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" of this harness = one pass of the hot path (T filter timesteps, full PSMF filter:
predict / y_hat = C mu_bar / innovation / r x r solve / low-rank updates of C, mu, P, V) over a
synthetic series that is already resident in HBM.  value = K * T / elapsed  (timesteps per
second of the whole job).  For N > 1 the d rows of C and y are sharded over the ranks
(strong scaling; blocked engine: one RCCL all-reduce of a 128 x 64 cross-Gram per block of 64 - r
timesteps, off the critical path; per-step engine: r+1 doubles per timestep); launch with
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` -- torch is used
only as rendezvous plumbing (gloo: id broadcast, barrier, max over ranks), never for compute.

Extra objects on the JSON line:
  roofline      dominant kernel: algorithmic bytes 8 d (r+1) per timestep x timesteps per launch / its
                average duration measured with HIP events on the library's stream, vs 8 TB/s HBM.
  cpu_baseline  the CPU oracle (numpy restatement of the reference algorithm, O(d r^2) form)
                timed on a bounded prefix of the same series on this box's host cores.
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md); measured copy peak is ~6.3 TB/s


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--d", type=int, default=100_000)
    ap.add_argument("--r", type=int, default=32)
    ap.add_argument("--T", type=int, default=10_000)
    ap.add_argument("--robust", type=int, default=0)
    ap.add_argument("--storage", default="f32")
    ap.add_argument("--cpu-steps", type=int, default=300, help="timesteps of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-y-pred", action="store_true")
    ap.add_argument("--workgroups", type=int, default=0)
    ap.add_argument("--engine", default="auto", choices=["auto", "step", "block"])
    return ap.parse_args()


class Series:
    """Synthetic series of ExperimentSynthetic/data.py semantics, generated shard-by-shard in
    time chunks so that neither the host nor PCIe ever holds the whole (T, d) array."""

    def __init__(self, d, r, T, seed, row0, d_local, robust):
        rng = np.random.default_rng(seed)
        C_true = rng.standard_normal((d, r))
        self.Ct = np.ascontiguousarray(C_true[row0:row0 + d_local].T)
        theta = 1e-3 * np.arange(1, r + 1)
        x = rng.standard_normal(r)
        self.X = np.empty((T, r))
        for t in range(1, T + 1):
            x = np.cos(2.0 * np.pi * theta * t + x)
            self.X[t - 1] = x
        self.T, self.d_local, self.robust = T, d_local, robust
        self.noise_seed = seed * 1000 + row0

    def chunks(self, chunk=500):
        rng = np.random.default_rng(self.noise_seed)
        sd = np.sqrt(0.1)
        for a in range(0, self.T, chunk):
            b = min(self.T, a + chunk)
            if self.robust:
                eps = rng.standard_t(3.0, (b - a, self.d_local)).astype(np.float32)
            else:
                eps = rng.standard_normal((b - a, self.d_local), dtype=np.float32)
            Y = (self.X[a:b] @ self.Ct).astype(np.float32)
            Y += np.float32(sd) * eps
            yield a, Y


def geo_engine_block(f):
    return f.geometry()["engine"] == "block"


def init_state(d, r, seed):
    rng = np.random.default_rng(seed + 7)
    C0 = (0.1 * rng.standard_normal((d, r))).astype(np.float32).astype(np.float64)
    return dict(C=C0, V=0.1 * np.eye(r), P=np.eye(r), Q=0.1 * np.eye(r), mu=np.zeros(r), rho=1.0, lam=1.8)


def cpu_baseline(args, series, st0):
    """Oracle (kind 'port') on the first cpu-steps timesteps; also returns its final state so the
    GPU result can be checked against it in the same run."""
    from oracle import psmf_oracle as O

    n = min(args.cpu_steps, args.T)
    Y = np.vstack([Yc for _, Yc in series.chunks(chunk=n)][:1])[:n].astype(np.float64)
    st = O.State(C=st0["C"].copy(), V=st0["V"].copy(), mu=st0["mu"].copy(), P=st0["P"].copy(), Q=st0["Q"].copy(),
                 rho=st0["rho"], lam=st0["lam"])
    mode = O.Mode(robust=bool(args.robust))
    t0 = time.perf_counter()
    st, _, _ = O.run_epoch(st, Y, mode, O.RandomWalkDyn(), want_grad=False)
    dt = time.perf_counter() - t0
    try:
        import threadpoolctl

        threads = max([p.get("num_threads", 1) for p in threadpoolctl.threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    info = dict(value=n / dt, unit="timesteps/s", cores=int(threads), kind="port",
                sample=f"first {n} of {args.T} timesteps of the same series (d={args.d}, r={args.r}), numpy float64 "
                       f"O(d r^2) restatement of the reference algorithm, {dt:.1f} s")
    return info, st, n


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    dist = None
    if world > 1:
        import torch.distributed as dist  # rendezvous plumbing only (gloo, CPU tensors)

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    from rpsmf_amd import _capi
    from rpsmf_amd.sharding import shard_rows

    d, r, T = args.d, args.r, args.T
    row0, d_local = shard_rows(d, world, rank)
    seed = 35833 if args.robust else 35853  # Makefile:55,64 of the reference
    series = Series(d, r, T, seed, row0, d_local, bool(args.robust))
    st0 = init_state(d, r, seed)

    f = _capi.DeviceFilter(d, r, robust=bool(args.robust), storage=args.storage, store_y_pred=not args.no_y_pred,
                           device=local_rank, row0=row0, d_local=d_local, n_workgroups=args.workgroups, engine=args.engine)
    if world > 1:
        import torch

        if rank == 0:
            uid = np.frombuffer(_capi.DeviceFilter.comm_unique_id(), dtype=np.uint8).copy()
        else:
            uid = np.zeros(_capi.UNIQUE_ID_BYTES, dtype=np.uint8)
        t = torch.from_numpy(uid)
        dist.broadcast(t, 0)
        f.comm_init(world, rank, t.numpy().tobytes())

    for a, Yc in series.chunks():
        f.upload_series(Yc, t0=a, T_total=T)

    def reset():
        f.set_state(st0["C"][row0:row0 + d_local], st0["V"], st0["P"], st0["Q"], st0["mu"], rho=st0["rho"],
                    lambda0=st0["lam"])

    def barrier():
        if dist is not None:
            dist.barrier()

    # ---- CPU baseline + in-run parity of the GPU path against it (rank 0, N = 1 only)
    cpu = None
    parity = None
    if world == 1 and args.cpu_steps > 0:
        cpu, st_cpu, n_cpu = cpu_baseline(args, series, st0)
        reset()
        f.run(0, n_cpu)
        s = f.get_state()
        rel = lambda a, b: float(np.max(np.abs(a - b)) / np.max(np.abs(b)))
        parity = dict(steps=n_cpu, C=rel(s["C"], st_cpu.C), V=rel(s["V"], st_cpu.V), mu=rel(s["mu"], st_cpu.mu),
                      P=rel(s["P"], st_cpu.P))

    # ---- untimed pre-warm: the HIP runtime grows its signal / kernel-argument pools the first time a whole pass worth of
    # launches is queued ahead of the GPU (a one-off ~80 ms stall in the second pass, tools/probe_stall.py)
    reset()
    for _ in range(2):
        f.run(0, T, sync=False)
    f.sync()
    # ---- timed region: K passes of T timesteps, state carried from pass to pass
    reset()
    for _ in range(args.warmup):
        f.run(0, T, sync=False)
    f.sync()
    if geo_engine_block(f):
        f.counters(reset=True)
        f.filter_kernel_time(reset=True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        f.run(0, T, sync=False)
    f.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    insitu = f.counters() if geo_engine_block(f) else None      # device-timer durations of the filter kernels of the timed region
    chained = f.filter_kernel_time() if geo_engine_block(f) else (0, 0.0)   # HIP events around the chained filter launches of the timed region
    if dist is not None:
        import torch

        te = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    value = args.steps * T / elapsed

    # ---- roofline of the dominant kernel, HIP events on the library's stream (psmf_time_kernel)
    bytes_per_step = (8.0 if args.storage == "f32" else 16.0) * d_local * (r + 1)   # read C, write C, read y, write y_hat
    geo = f.geometry()
    traffic = None
    if geo["engine"] == "block":
        # one launch of the coefficient-space filter kernel = one block of B timesteps; the two d-sized
        # products of the block (cross-Gram, apply) run concurrently on a second stream
        B = geo["block_steps"]
        t_filter = f.time_kernel(0, 20)
        t_gram = f.time_kernel(1, 50)
        t_apply = f.time_kernel(2, 50)
        # Duration of the dominant kernel over the TIMED REGION.  Chained blocks (the default): one launch of
        # psmf_blk_filter3 per pass, bracketed by HIP events on the stream it runs on (psmf_filter_kernel_time); it advances
        # every timestep of the pass.  Unchained (PSMF_BLOCK_CHAIN=0): one launch per block, timed by the kernel's own
        # s_memrealtime stamps (HIP events between those launches add 5-8 us each).  The stand-alone HIP-event figure of a
        # single block (psmf_time_kernel) is reported beside it.
        t_block_insitu = insitu["filter_us_mean"] if insitu and insitu["filter_launches"] > 0 else t_filter
        if chained[0] > 0:
            kernel, kernel_us, steps_per_launch = "psmf_blk_filter3", 1e3 * chained[1] / chained[0], float(T)   # (at most 16 launches between two syncs are timed)
        else:
            full_blocks_only = (T % B == 0)
            kernel, kernel_us = "psmf_blk_filter3", t_block_insitu
            steps_per_launch = B if full_blocks_only else T / (insitu["filter_launches"] / args.steps)
        zbytes = (4.0 if args.storage == "f32" else 8.0) * d_local * 64
        pmc = os.path.join(ROOT, "profiles", "r1_pmc_traffic_block_engine.json")
        if (d, r, args.storage, world) == (100_000, 32, "f32", 1) and os.path.exists(pmc):
            # rocprofv3 --pmc passes of this workload, taken with PSMF_BLOCK_CHAIN=0 (counter collection serialises the
            # kernels, which a chained launch waiting for the bulk stream cannot survive): bytes per block of B timesteps,
            # scaled to the blocks one launch advances
            traffic = json.load(open(pmc))["traffic_bytes_per_launch"] * (steps_per_launch / B)
        extra = {"steps_per_launch": steps_per_launch,
                 "blocks_per_launch": steps_per_launch / B,
                 "block_us_in_kernel": t_block_insitu,
                 "one_block_us_hip_events_standalone": t_filter,
                 "gap_between_blocks_us": insitu["filter_gap_us_mean"] if insitu else None,
                 "kernels_us": {"psmf_blk_filter3": t_filter, "psmf_blk_xgram2+xreduce2": t_gram, "psmf_blk_apply2": t_apply},
                 "bulk_kernels_GBps": {"cross-Gram (reads Z and the next series block)": 1.5 * zbytes / (t_gram * 1e-6) / 1e9,
                                       "apply (reads Z, writes C and y_hat)": 2 * zbytes / (t_apply * 1e-6) / 1e9},
                 "note": "blocked engine: the filter kernel is a latency-bound chain of r x r stages (one workgroup, f64-MFMA Newton-Schulz), "
                         "one launch per pass; achieved = step-at-a-time algorithmic bytes of the timesteps it advances / its duration "
                         "(HIP events on its stream)"}
    else:
        kernel, kernel_us, steps_per_launch = "psmf_sweep_solve", f.time_kernel(0, 300), 1
        extra = {"steps_per_launch": 1, "kernels_us": {"psmf_sweep_solve": kernel_us, "psmf_serial": f.time_kernel(1, 300)}}
        pmc = os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")
        if (d, r, args.storage, world) == (100_000, 32, "f32", 1) and os.path.exists(pmc):
            traffic = json.load(open(pmc))["traffic_bytes_per_launch"]   # rocprofv3 --pmc passes of this workload
    alg_bytes = bytes_per_step * steps_per_launch
    achieved = alg_bytes / (kernel_us * 1e-6) / 1e9
    if rank == 0:
        line = {
            "metric": "PSMF filter timesteps/sec at d=100k r=32" if (d, r) == (100_000, 32) else f"PSMF filter timesteps/sec at d={d} r={r}",
            "value": value,
            "unit": "timesteps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",      # the arithmetic type: float64 accumulation of every d -> r contraction and float64 r x r state
            "storage_dtype": args.storage,   # C, y, y_hat in HBM (f32: one rounding of C per block)
            "data": "synthetic",
            "config": {"workload": f"{'rPSMF' if args.robust else 'PSMF'} full filter, random-walk dynamics, d={d} r={r} "
                                   f"T={T} synthetic Gaussian series, rows sharded over {world} GPU(s)",
                       "d": d, "r": r, "T": T, "timesteps_per_pass": T, "store_y_pred": not args.no_y_pred,
                       "us_per_timestep": 1e6 * elapsed / (args.steps * T), "engine": geo["engine"], "geometry": geo},
            "roofline": dict({"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": kernel, "kernel_us": kernel_us,
                              "algorithmic_bytes_per_launch": alg_bytes,
                              "whole_job_frac": value * 8.0 * d * (r + 1) / (world * HBM_PEAK_GBS * 1e9)}, **extra),
        }
        if cpu is not None:
            line["cpu_baseline"] = cpu
        if parity is not None:
            line["parity_vs_cpu_oracle"] = parity
        print(json.dumps(line), flush=True)
    f.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
