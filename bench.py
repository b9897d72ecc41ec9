#!/usr/bin/env python3
"""Benchmark of the Gibbs-sweep hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

A "step" is one sweep of the hot path over every variable of the workload graph (one epoch of
gibbsthread, numbskull/inference.py:10-33; for the *_learn workloads one epoch of learnthread,
learning.py:12-125).  The default workload is the one BASELINE.json's metric is quoted on: the
2500x4000 (10M-variable) 2D Ising grid of ising/ising.cpp:134-199, binary variables, EQUAL
factors, one fixed weight 0.1, inference only, inputs resident in HBM before the timed region.

Prints ONE JSON line (rank 0): metric = variable-updates/sec over all GPUs, plus
  roofline     algorithmic bytes per launch (106.9 B/update on the grid) /
               average launch duration measured with HIP events on the library's stream
  cpu_baseline the CPU restatement of the reference algorithm (oracle/, Hogwild threads like the
               reference's run_pool) timed on this node's host cores on a bounded sample.

N > 1 (launched by torch.distributed.run, one rank per GPU): the grid is range-partitioned by
variable id with the reference's shard formula and the owned value slices are all-gathered over
RCCL after every sweep (numbskull_amd/distributed.py); total work is fixed => "strong" scaling.
"""

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import numpy as np          # noqa: E402

WORKLOADS = {
    # name: (rows, cols, learning)
    "ising10m": (2500, 4000, False),          # BASELINE configs[2]/[3] inference (metric config)
    "ising1m": (1000, 1000, False),           # BASELINE configs[1]
    "ising10m_learn": (2500, 4000, True),     # BASELINE configs[2] learning half
    "ising1m_learn": (1000, 1000, True),
    # 4M boolean variables, ISTRUE / OR / EQUAL factors of arity 1..3 with one weight per factor
    # (the shape of feature-weighted DeepDive graphs): exercises the per-lane-weight shape tiles
    "boolw4m": (2000, 2000, False),
    "boolw4m_learn": (2000, 2000, True),
    # scaled-down BASELINE configs[4]: mixed-arity LR graph (25 % categorical variables, ISTRUE / OR /
    # IMPLY_MLN / OR_CAT / IMPLY_MLN_CAT / AND_CAT factors, 10^5 weights), inference and learning
    "lr5m": (2500, 2000, False),
    "lr5m_learn": (2500, 2000, True),
}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8 TB/s HBM3E peak


def build_graph(rows, cols, learning, seed=20240602, name=None):
    from numbskull_amd import graphgen
    if name in ("boolw4m", "boolw4m_learn"):
        g = graphgen.boolean_weighted_graph(rows * cols, seed=seed)
        if learning:                     # free weights, half of the variables evidence
            rng = np.random.Generator(np.random.PCG64(seed + 1))
            g[0]["isFixed"] = False
            g[0]["initialValue"] = 0.0
            g[1]["isEvidence"] = rng.random(len(g[1])) < 0.5
            g[1]["initialValue"] = rng.integers(0, 2, len(g[1]))
        return g
    if name in ("lr5m", "lr5m_learn"):
        return graphgen.mixed_lr_graph(rows * cols, seed=20240603)
    if not learning:
        return graphgen.ising_grid(rows, cols, weight=0.1, fixed=True)
    # learning variant: two free weights, every variable evidence;
    # the evidence configuration is a seeded random field (the planted-weight recovery run is in
    # tests/ and DESIGN.md; throughput does not depend on the configuration)
    rng = np.random.Generator(np.random.PCG64(seed))
    ev = rng.integers(0, 2, rows * cols)
    return graphgen.ising_grid(rows, cols, weight=0.0, fixed=False, two_weights=True, evidence=ev)


def cpu_baseline(fg, learning, budget_s=20.0):
    """Oracle (CPU restatement of the reference algorithm) on this node's cores, bounded sample."""
    from oracle import binding as orc
    og = orc.Graph(fg.weight, fg.variable, fg.factor, fg.fmap, fg.vmap, fg.factor_index)
    nvar = len(fg.variable)
    cores = os.cpu_count() or 1
    vv, ve, wv, cnt = og.initial_state()
    run = (lambda n: og.learn_hogwild(cores, n, vv, ve, wv, 1e-7, 0.95, 2, 0.01, 1, False, 1)) \
        if learning else (lambda n: og.gibbs_hogwild(cores, n, vv, wv, cnt, 1, True, False))
    t0 = time.time()
    rc = run(1)
    t1 = time.time() - t0
    assert rc == 0
    total_t, total_n = t1, 1
    while total_t < budget_s * 0.6:            # a bounded sample: ~10-20 s of CPU work
        extra = int(max(1, min(200, (budget_s * 0.75 - total_t) // max(total_t / total_n, 1e-3))))
        t0 = time.time()
        run(extra)
        total_t += time.time() - t0
        total_n += extra
    # one thread = the reference's own sequential scan (SURVEY.md section 8d asks for T = 1 too)
    run1 = (lambda n: og.learn_hogwild(1, n, vv, ve, wv, 1e-7, 0.95, 2, 0.01, 1, False, 1)) \
        if learning else (lambda n: og.gibbs_hogwild(1, n, vv, wv, cnt, 1, True, False))
    t0 = time.time()
    assert run1(1) == 0
    single = nvar / (time.time() - t0)
    return {"value": nvar * total_n / total_t, "unit": "variable-updates/s", "cores": cores,
            "kind": "port", "single_thread": single,
            "sample": "%d sweep(s) of the same %d-variable grid, %d Hogwild threads "
                      "(reference shard formula), %.1f s" % (total_n, nvar, cores, total_t)}


def side_run(name, seed, steps, warmup):
    """Secondary measurements on one GPU (BASELINE configs[1]: the 1M-variable grid; configs[2]'s
    learning half: the 10M grid with two free weights), reported beside the headline line under
    "also"."""
    import ctypes as C
    import io
    from contextlib import redirect_stdout
    import torch
    import numbskull_amd
    from numbskull_amd import _lib
    rows, cols, learning = WORKLOADS[name]
    w, v, f, fm, dm, edges = build_graph(rows, cols, learning)
    ns = numbskull_amd.NumbSkull(quiet=True, seed=seed)
    with redirect_stdout(io.StringIO()):
        ns.loadFactorGraph(w, v, f, fm, dm, int(edges))
    fg = ns.factorGraphs[0]
    L, h = _lib.lib(), fg._engine()
    info = fg.info()

    def run(n):
        if learning:                     # config #3 parameters: step 1e-7, L2 0.01
            _lib.check(L.nsk_learn_sweeps(h, n, 1e-7, 1.0, 2, 0.01, 1, 0))
        else:
            _lib.check(L.nsk_gibbs_sweeps(h, n, 1, 0))
    run(warmup)
    torch.cuda.synchronize()
    _lib.check(L.nsk_profile_begin(h))
    t0 = time.perf_counter()
    run(steps)
    ms, nl = C.c_double(), C.c_int64()
    _lib.check(L.nsk_profile_end(h, C.byref(ms), C.byref(nl)))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    fg.close()
    alg = info["layout_bytes_learning" if learning else "layout_bytes_inference"] * steps / (ms.value / 1e3) / 1e9
    return {"value": rows * cols * steps / dt, "unit": "variable-updates/s", "steps": steps,
            "ms_per_step": dt * 1e3 / steps, "roofline_frac": alg / HBM_PEAK_GBS,
            "avg_launch_us": ms.value * 1e3 / max(1, nl.value)}


def dominant_kernel(workload, learning, info):
    """Name of the kernel family the launch average is dominated by (profiles/*_kernel_stats.csv)."""
    if not info["nfast"]:
        return "k_learn_phase" if learning else "k_gibbs_phase"
    if workload.startswith("lr"):
        return "k_learn_general" if learning else "k_gibbs_general"
    if workload.startswith("boolw"):
        return "k_learn_fast+k_learn_general" if learning else "k_gibbs_fast+k_gibbs_general"
    return "k_learn_seg" if learning else "k_gibbs_seg"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="ising10m", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary 1M-grid measurement")
    ap.add_argument("--seed", type=int, default=20240601)
    args = ap.parse_args()

    import torch
    import numbskull_amd
    from numbskull_amd import _lib
    from numbskull_amd.distributed import PartitionedSampler, shard_range

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with that many ranks" % args.gpus)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # test hooks for a box with fewer GPUs than ranks: NSK_BENCH_ONE_DEVICE puts every rank on
        # device 0, NSK_BENCH_BACKEND=gloo replaces RCCL (the library then falls back to the
        # torch.distributed exchange loop); the driver's runs use neither
        if os.environ.get("NSK_BENCH_ONE_DEVICE"):
            local_rank = 0
        torch.cuda.set_device(local_rank)
        dist.init_process_group(os.environ.get("NSK_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)

    rows, cols, learning = WORKLOADS[args.workload]
    g = build_graph(rows, cols, learning, name=args.workload)
    nvar = rows * cols
    ns = numbskull_amd.NumbSkull(quiet=True, device=local_rank, seed=args.seed,
                                 head_by_vid=args.workload.startswith("lr"))
    own = shard_range(rank, world, nvar)
    w, v, f, fm, dm, edges = g
    import io
    from contextlib import redirect_stdout
    with redirect_stdout(io.StringIO()):
        ns.loadFactorGraph(w, v, f, fm, dm, int(edges), own_range=own if world > 1 else None)
    fg = ns.factorGraphs[0]
    L, h = _lib.lib(), fg._engine()
    info = fg.info()
    sampler = PartitionedSampler(fg, dist, torch, rank, world) if world > 1 else None
    lr = (1e-7, 0.95, 2, 0.01, 1)       # step, decay, L2, reg_param, truncation (config #3)
    if args.workload.startswith("lr") or args.workload.startswith("boolw"):
        lr = (1e-3, 0.95, 2, 0.01, 1)

    def run(n):
        if learning:
            if world > 1:
                sampler.learn(n, *lr)
            else:
                _lib.check(L.nsk_learn_sweeps(h, n, lr[0], 1.0, lr[2], lr[3], lr[4], 0))
        else:
            if world > 1:
                sampler.gibbs(n, True, False)
            else:
                _lib.check(L.nsk_gibbs_sweeps(h, n, 1, 0))

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run(args.warmup)
    fence()
    import ctypes as C
    _lib.check(L.nsk_profile_begin(h))
    t0 = time.perf_counter()
    run(args.steps)
    ms_ev, launches = C.c_double(), C.c_int64()
    _lib.check(L.nsk_profile_end(h, C.byref(ms_ev), C.byref(launches)))
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    copy_gbs = None
    if rank == 0:
        cg = C.c_double()
        if L.nsk_selftest_stream(local_rank, 4 << 30, 64, 5, C.byref(cg)) == 0:
            copy_gbs = cg.value           # achievable HBM copy rate on this GPU, same run (4 GiB,
                                          # non-temporal 16-byte accesses, 4 loads in flight per lane)
    if rank == 0:
        alg_sweep = info["alg_bytes_learning"] if learning else info["alg_bytes_inference"]
        # roofline.achieved: the bytes one launch must move in the compiled device layout (tile
        # words, position arrays, distinct values read, stores, tallies -- nsk_compile.cpp "layout
        # bytes") / the average launch duration.  The SURVEY 8(d) CSR-model figure is reported
        # beside it (alg_bytes_per_update_csr); the inlined layout moves far fewer bytes than that
        # model, so a fraction computed from it would exceed 1.
        lay_sweep = info["layout_bytes_learning"] if learning else info["layout_bytes_inference"]
        nlaunch = max(1, launches.value)
        lay_per_launch = lay_sweep * args.steps / nlaunch
        launch_s = (ms_ev.value / 1e3) / nlaunch
        achieved = lay_per_launch / launch_s / 1e9
        traffic = None
        tp = os.path.join(REPO, "profiles", "traffic.json")
        if os.path.exists(tp) and world == 1:      # PMC bytes were collected for the one-GPU launch
            try:
                traffic = json.load(open(tp)).get(args.workload)
            except Exception:
                traffic = None
        out = {
            "metric": "variable-updates/sec", "value": nvar * args.steps / dt,
            "unit": "variable-updates/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": ("%d boolean variables, %d ISTRUE/OR/EQUAL factors with one weight each, "
                                    "inference only, chromatic scan, seed %d" % (nvar, len(f), args.seed))
                       if args.workload.startswith("boolw4m") else
                       ("mixed-arity LR graph: %d variables (25%% categorical), %d factors, %d weights, %s"
                        % (nvar, len(f), len(w), "learning" if learning else "inference"))
                       if args.workload.startswith("lr") else
                       "%dx%d Ising grid (%d binary variables, %d EQUAL factors), %s, "
                       "chromatic scan, seed %d"
                       % (rows, cols, nvar, len(f), "learning (2 free weights, L2)"
                          if learning else "inference only, weight 0.1 fixed", args.seed),
                       "name": args.workload, "partition": "range by variable id, %d shard(s)" % world,
                       "colors": info["ncolors"], "value_bytes": info["value_bytes"]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "actual_GBs": (traffic / launch_s / 1e9) if traffic else None,
                         "layout_bytes_per_update": lay_sweep * world / nvar,
                         "alg_bytes_per_update_csr": alg_sweep * world / nvar,
                         "csr_model_GBs": alg_sweep * args.steps / (ms_ev.value / 1e3) / 1e9,
                         "kernel": dominant_kernel(args.workload, learning, info),
                         "stream_copy_GBs": copy_gbs,
                         "launches": nlaunch, "avg_launch_us": launch_s * 1e6},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(fg, learning)
        if world == 1 and args.workload == "ising10m" and not args.no_extra:
            out["also"] = {"ising1m": side_run("ising1m", args.seed, 1000, 100),
                           "ising10m_learn": side_run("ising10m_learn", args.seed, 100, 10)}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
