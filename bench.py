#!/usr/bin/env python3
"""Benchmark of the Gibbs-sweep hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

A "step" is one sweep of the hot path over every variable of the workload graph (one epoch of
gibbsthread, numbskull/inference.py:10-33; for the *_learn workloads one epoch of learnthread,
learning.py:12-125).  The default workload is the one BASELINE.json's metric is quoted on: the
2500x4000 (10M-variable) 2D Ising grid of ising/ising.cpp:134-199, binary variables, EQUAL
factors, one fixed weight 0.1, inference only, inputs resident in HBM before the timed region.

Prints ONE JSON line (rank 0): metric = variable-updates/sec over all GPUs, plus
  roofline     bytes one launch must move in the compiled device layout (nsk_compile.cpp "layout
               bytes") / average launch duration measured with HIP events on the library's
               stream; the SURVEY.md section 8d CSR-model figure rides along as
               alg_bytes_per_update_csr
  cpu_baseline the CPU restatement of the reference algorithm (oracle/, Hogwild threads like the
               reference's run_pool) timed on this node's host cores on a bounded sample
  parity       invariants of the state after the timed region (tally bounds, mean marginal) and,
               with the CPU baseline, the nearest-neighbour agreement statistic of both runs.

N > 1: one rank per GPU (torch.distributed.run; started by this script as a child process when
it is not already running under it), the grid range-partitioned by variable id with the
reference's shard formula, boundary values all-gathered over RCCL after every sweep
(numbskull_amd/distributed.py); total work is fixed => "strong" scaling.
"""

import argparse
import json
import os
import subprocess
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
    # BASELINE configs[4]: mixed-arity LR graph (25 % categorical variables, ISTRUE / OR /
    # IMPLY_MLN / OR_CAT / IMPLY_MLN_CAT / AND_CAT factors), scaled down 10x and at full size
    "lr5m": (2500, 2000, False),
    "lr5m_learn": (2500, 2000, True),
    "lr50m": (10000, 5000, False),
    "lr50m_learn": (10000, 5000, True),
    "ising64k": (256, 256, False),            # plumbing tests
    "ising256k": (512, 512, False),           # (where wide quads start to pay: tools/sessions/r6_s25.sh)
    "ising500k": (500, 1000, False),
    "lr300k_learn": (600, 500, True),
    # 4x / 10x the metric config.  With implicit adjacency a sweep of the 40M grid moves ~190 MB (it was
    # 800 MB in round 2) and fits the 256 MiB Infinity Cache again; the 100M grid (~450 MB per sweep) is
    # the one beyond it
    "ising4m": (2000, 2000, False),
    "ising40m": (5000, 8000, False),
    "ising4m_learn": (2000, 2000, True),
    "ising40m_learn": (5000, 8000, True),
    "ising100m": (10000, 10000, False),
}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8 TB/s HBM3E peak
# 256 MiB Infinity Cache (memory-side): a sweep whose streams fit in it is served partly from there
# from the second sweep on, and FETCH_SIZE counts those hits as fetches (MI355X_MICROARCH.md).  The
# 10M grid's sweep moves ~48 MB and the 40M grid's ~190 MB: both fit, so their fractions of the HBM peak
# are upper bounds on what HBM itself delivered (`stream_fits_infinity_cache` says so per workload).
INFINITY_CACHE_BYTES = 256 << 20


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
    if name is not None and name.startswith("lr"):
        return graphgen.mixed_lr_graph(rows * cols, seed=20240603)     # 10^5 weights at 5M, 10^6 at 50M
    if not learning:
        return graphgen.ising_grid(rows, cols, weight=0.1, fixed=True)
    # learning variant: two free weights, every variable evidence;
    # the evidence configuration is a seeded random field (the planted-weight recovery run is in
    # tests/ and DESIGN.md; throughput does not depend on the configuration)
    rng = np.random.Generator(np.random.PCG64(seed))
    ev = rng.integers(0, 2, rows * cols)
    return graphgen.ising_grid(rows, cols, weight=0.0, fixed=False, two_weights=True, evidence=ev)


def build_shard(rows, cols, learning, name, lo, hi):
    """Rank-local graph of an N-rank run: (graph, global_ids, own_local, nfactor_total, nweight_total,
    seconds).  The LR workloads generate the shard alone (graphgen.mixed_lr_shard: 4 bytes per variable
    of the whole graph + this rank's factors -- the reference's minions load only their partition,
    salt/src/numbskull_minion.py:185); the grids are cut out of the generated graph."""
    from numbskull_amd import graphgen
    t0 = time.time()
    if name.startswith("lr"):
        g, gids, own_local = graphgen.mixed_lr_shard(rows * cols, lo, hi, seed=20240603)
        return g, gids, own_local, None, len(g[0]), time.time() - t0
    if not learning:
        # the grid's shard from its own rows (graphgen.ising_grid_shard == extract_shard of the whole grid): a rank of the
        # 100M grid holds 12.5M cells, not 100M
        g, gids, own_local = graphgen.ising_grid_shard(rows, cols, lo, hi, weight=0.1, fixed=True)
        nf = (rows - 1) * cols + rows * (cols - 1)
        return g, gids, own_local, nf, len(g[0]), time.time() - t0
    whole = build_graph(rows, cols, learning, name=name)
    nf, nw = len(whole[2]), len(whole[0])
    g, gids, own_local = graphgen.extract_shard(whole, lo, hi)
    return g, gids, own_local, nf, nw, time.time() - t0


def peak_rss_gb():
    import resource
    return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2.0 ** 20


def native_oracle():
    """The CPU baseline is built ON THIS NODE with -O3 -march=native; the
    portable build in oracle/ (made in the build container, -O2) is the fallback."""
    from oracle import binding as orc
    src = os.path.join(REPO, "oracle", "nsk_oracle.c")
    out_dir = os.path.join(REPO, "oracle", "_native")
    out = os.path.join(out_dir, "libnsk_oracle_native.so")
    flags = "-O3 -march=native -ffp-contract=off"
    try:
        os.makedirs(out_dir, exist_ok=True)
        subprocess.check_call(["gcc"] + flags.split() + ["-fPIC", "-std=c99", "-D_GNU_SOURCE", "-shared",
                                                         "-o", out, src, "-lm", "-lpthread"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        orc.use_library(out)
        return flags
    except Exception:
        return "-O2 -mfma -ffp-contract=off (portable build; native rebuild failed)"


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def grid_agreement(values, rows, cols):
    """Fraction of grid edges whose two variables agree (the sufficient statistic of the EQUAL
    pair model: its expectation is a function of the weight alone)."""
    x = np.asarray(values).reshape(rows, cols)
    same = int((x[1:, :] == x[:-1, :]).sum()) + int((x[:, 1:] == x[:, :-1]).sum())
    return same / float((rows - 1) * cols + rows * (cols - 1))


def cpu_baseline(fg, learning, budget_s=20.0, grid=None, head_by_vid=False, lr=(1e-7, 0.95, 2, 0.01, 1)):
    """Oracle (CPU restatement of the reference algorithm) on this node's cores, bounded sample:
    whole sweeps of the SAME graph until ~60 % of `budget_s` is spent (one sweep at least)."""
    flags = native_oracle()
    from oracle import binding as orc
    og = orc.Graph(fg.weight, fg.variable, fg.factor, fg.fmap, fg.vmap, fg.factor_index,
                   head_by_vid=head_by_vid)
    nvar = len(fg.variable)
    cores = os.cpu_count() or 1
    vv, ve, wv, cnt = og.initial_state()
    run = (lambda n: og.learn_hogwild(cores, n, vv, ve, wv, lr[0], lr[1], lr[2], lr[3], lr[4], False, 1)) \
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
    agree = grid_agreement(vv, *grid) if grid and not learning else None
    mean_marg = float(cnt.sum()) / (nvar * total_n) if not learning else None
    # one thread = the reference's own sequential scan (SURVEY.md section 8d asks for T = 1 too)
    vv1, ve1, wv1, cnt1 = og.initial_state()
    run1 = (lambda n: og.learn_hogwild(1, n, vv1, ve1, wv1, lr[0], lr[1], lr[2], lr[3], lr[4], False, 1)) \
        if learning else (lambda n: og.gibbs_hogwild(1, n, vv1, wv1, cnt1, 1, True, False))
    single = None               # None in the line = not measured (skipped for graphs beyond 12M variables)
    if nvar <= 12_000_000:      # (a one-thread sweep of the 50M graph alone would take the whole budget)
        t0 = time.time()
        assert run1(1) == 0
        single = nvar / (time.time() - t0)
    # the baseline is the BETTER of the two thread counts (two shared weights bouncing between 256 cores make
    # the all-core learning run slower than one thread): both are listed
    allcores = nvar * total_n / total_t
    best_single = single is not None and single > allcores
    return {"value": single if best_single else allcores, "unit": "variable-updates/s",
            "cores": 1 if best_single else cores,
            "kind": "port", "single_thread": single, "all_cores": allcores, "all_cores_threads": cores,
            "single_thread_note": None if single is not None else "skipped: one sweep of this graph on one thread exceeds the sample budget",
            "cflags": flags, "cpu": cpu_model(),
            "sample": "%d sweep(s) of the same %d-variable graph, %d Hogwild threads "
                      "(reference shard formula), %.1f s%s; value = the faster of the two" % (
                          total_n, nvar, cores, total_t, "" if single is None else "; 1 sweep on one thread"),
            "_agreement": agree, "_mean_marginal": mean_marg, "_sweeps": total_n}


def state_checks(fg, info, tallied_sweeps, grid, learning):
    """Cheap invariants of the device state after the timed region (downloaded once)."""
    fg._pull(0, 0, values=True, weights=True, count=True)
    own = fg.own_range if fg.own_range is not None else (0, len(fg.variable))
    v = fg.var_value[0][own[0]:own[1]]
    card = fg.variable["cardinality"][own[0]:own[1]]
    out = {"values_in_domain": bool(((v >= 0) & (v < card)).all())}
    if not learning:
        cs = fg.cstart
        cnt = fg.count[cs[own[0]]:cs[own[1]]]
        out["tally_min"] = int(cnt.min()) if len(cnt) else 0
        out["tally_max"] = int(cnt.max()) if len(cnt) else 0
        out["tallied_sweeps"] = int(tallied_sweeps)
        out["tally_in_bounds"] = bool(out["tally_min"] >= 0 and out["tally_max"] <= tallied_sweeps)
        if grid:
            out["mean_marginal"] = float(cnt.sum()) / max(1, len(cnt) * tallied_sweeps)
            out["mean_marginal_ok"] = bool(abs(out["mean_marginal"] - 0.5) < 0.01)
            if fg.own_range is None:
                out["edge_agreement"] = grid_agreement(fg.var_value[0], *grid)
    else:
        w = fg.weight_value[0]
        out["weights_finite"] = bool(np.isfinite(w).all())
        out["weights"] = [float(x) for x in w[:4]]
    return out


REPEATS = 5      # timed blocks of K steps each; the line's value is the MEDIAN block


def timed_blocks(run, fence, L, h, steps, reps):
    """`reps` blocks of exactly `steps` sweeps, each bracketed by fence() (barrier + device
    synchronise) on both sides and by HIP events on the library's stream.  Returns the host
    wall time, event time and launch count of every block."""
    import ctypes as C
    from numbskull_amd import _lib
    out = []
    for _ in range(reps):
        fence()
        _lib.check(L.nsk_profile_begin(h))
        t0 = time.perf_counter()
        run(steps)
        _lib.check(L.nsk_profile_mark(h))          # closing event: recorded, not waited for
        fence()
        dt = time.perf_counter() - t0
        ms, nl = C.c_double(), C.c_int64()
        _lib.check(L.nsk_profile_read(h, C.byref(ms), C.byref(nl)))    # (after the clock has stopped)
        out.append((dt, ms.value, nl.value))
    return out


def spread(blocks, steps):
    dts = sorted(b[0] for b in blocks)
    med = dts[len(dts) // 2]
    return {"n": len(dts), "steps_each": steps, "ms_per_step_min": dts[0] * 1e3 / steps,
            "ms_per_step_median": med * 1e3 / steps, "ms_per_step_max": dts[-1] * 1e3 / steps,
            "rel_spread": (dts[-1] - dts[0]) / med}


def median_block(blocks):
    return sorted(blocks)[len(blocks) // 2]


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
    is_lr = name.startswith("lr")
    t_gen = time.time()
    w, v, f, fm, dm, edges = build_graph(rows, cols, learning, name=name)
    t_gen = time.time() - t_gen
    # (the LR generator's IMPLY_MLN / IMPLY_MLN_CAT factors use the intended head lookup fmap[l].vid: the reference's
    #  literal indexing is out of bounds on this graph -- SURVEY.md section 8d, config #5)
    ns = numbskull_amd.NumbSkull(quiet=True, seed=seed, head_by_vid=is_lr)
    t_load = time.time()
    with redirect_stdout(io.StringIO()):
        ns.loadFactorGraph(w, v, f, fm, dm, int(edges))
    t_load = time.time() - t_load
    fg = ns.factorGraphs[0]
    L, h = _lib.lib(), fg._engine()
    info = fg.info()
    step = 1e-3 if is_lr else 1e-7       # config #3: step 1e-7, L2 0.01; the LR graphs: 1e-3

    def run(n, burnin=0):
        if learning:
            _lib.check(L.nsk_learn_sweeps(h, n, step, 1.0, 2, 0.01, 1, 0))
        else:
            _lib.check(L.nsk_gibbs_sweeps(h, n, 1, burnin))
    run(warmup, 1)                       # warm-up = burn-in: the tallies start after the transient
    blocks = timed_blocks(run, torch.cuda.synchronize, L, h, steps, REPEATS)
    dt, ms, nl = median_block(blocks)
    checks = state_checks(fg, info, steps * REPEATS, (rows, cols) if not is_lr else None, learning)
    clipped = fg.info()["learn_clipped"] if learning else None
    fg.close()
    lay_sweep = info["layout_bytes_learning" if learning else "layout_bytes_inference"]
    lay = lay_sweep * steps / (ms / 1e3) / 1e9
    out = {"value": rows * cols * steps / dt, "unit": "variable-updates/s", "steps": steps,
           "ms_per_step": dt * 1e3 / steps, "repeats": spread(blocks, steps), "roofline_frac": lay / HBM_PEAK_GBS,
           "sweep_stream_bytes": lay_sweep, "stream_fits_infinity_cache": bool(lay_sweep < INFINITY_CACHE_BYTES),
           "avg_launch_us": ms * 1e3 / max(1, nl), "parity": checks}
    if learning:
        out["learn_clipped"] = clipped   # weight updates whose step the per-class cap shrank
    if is_lr:
        out["config"] = {"workload": "mixed-arity LR graph: %d variables (25%% categorical), %d factors, %d weights, %s; "
                                     "head_by_vid (SURVEY.md section 8d: the literal head index is out of bounds on this graph)"
                                     % (rows * cols, len(f), len(w), "learning" if learning else "inference"),
                         "generate_s": round(t_gen, 2), "load_and_compile_s": round(t_load, 2),
                         "compile_s": round(info["compile_seconds"], 2), "colors": info["ncolors"]}
        tr = profile_record("traffic.json", name)
        if tr:                           # counter bytes per class launch / layout bytes per class launch (profiles/)
            out["traffic_per_launch"] = tr
            out["traffic_over_layout"] = tr / (lay_sweep * steps / max(1, nl))
    return out


def multi_side_run(name, args, dist, torch, rank, world, local_rank, steps=20, warmup=5):
    """A second workload on the same N ranks (inference grids only): shard-local graphs, the same exchange ladder, K sweeps
    bracketed by barriers, the MAX over ranks; rank 0 gets the record."""
    import io
    from contextlib import redirect_stdout
    import numbskull_amd
    from numbskull_amd.distributed import PartitionedSampler, shard_range
    rows, cols, learning = WORKLOADS[name]
    nvar = rows * cols
    own = shard_range(rank, world, nvar)
    g, gids, own_local, _, _, t_gen = build_shard(rows, cols, learning, name, own[0], own[1])
    ns = numbskull_amd.NumbSkull(quiet=True, device=local_rank, seed=args.seed)
    with redirect_stdout(io.StringIO()):
        ns.loadFactorGraph(g[0], g[1], g[2], g[3], g[4], int(g[5]), own_range=own_local, global_ids=gids)
    fg = ns.factorGraphs[0]
    sampler = PartitionedSampler(fg, dist, torch, rank, world, nvar_global=nvar)
    sampler.gibbs(warmup, True, burnin=True)

    def fence():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
    times = []
    for _ in range(3):
        fence()
        t0 = time.time()
        sampler.gibbs(steps, True)
        sampler.check()
        fence()
        times.append(time.time() - t0)
    t = torch.tensor([sorted(times)[1]], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    rec = {"value": nvar * steps / float(t.item()), "unit": "variable-updates/s", "n_gpus": world, "steps": steps,
           "ms_per_step": float(t.item()) * 1e3 / steps, "exchange": sampler.ladder["rung"],
           "config": {"workload": "%dx%d Ising grid (%d binary variables), inference only, %d range shards generated per rank"
                                  % (rows, cols, nvar, world), "generate_s": round(t_gen, 2)}}
    fg.close()
    return rec


def profile_record(fname, workload):
    """A per-workload record of profiles/<fname> (tools/collect_profiles.sh), None when absent."""
    tp = os.path.join(REPO, "profiles", fname)
    try:
        return json.load(open(tp)).get(workload)
    except Exception:
        return None


def profiles_stale():
    """True when profiles/ was collected at another library than the one that runs: profiles/r6_COMMIT.txt names the
    library version string + source hash tools/collect_profiles.sh saw."""
    from numbskull_amd import _lib
    try:
        want = open(os.path.join(REPO, "profiles", "r6_COMMIT.txt")).read()
    except Exception:
        return True
    return library_fingerprint(_lib) not in want


def library_fingerprint(_lib):
    """nsk_version() + a hash of the kernel sources the in-tree library was built from."""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(REPO, "numbskull_amd", "csrc")
    for fn in sorted(os.listdir(src)):
        if fn.endswith((".h", ".hip", ".cpp")):
            h.update(open(os.path.join(src, fn), "rb").read())
    v = _lib.lib().nsk_version()
    return "%s src=%s" % (v.decode() if isinstance(v, bytes) else v, h.hexdigest()[:16])


def roofline_bound(kernel):
    """What binds the dominant kernel (from the stage ablations and PMC passes under
    profiles/): the draw-table kernels read almost no stream and are bound by instruction issue along a
    chain of dependent steps; the entry-parallel kernels by the latency of their dependent gathers at the
    occupancy their registers allow (no unit saturated); the CSR / exp-per-update kernels by HBM."""
    if kernel.endswith("_seg_tab") or kernel.endswith("_seg_tabw") or kernel.endswith("_seg_tab_p2p"):
        return "issue"
    if kernel.endswith("_ep") or kernel.endswith("_ep_w5") or kernel.endswith("_ep_w4"):
        return "latency"
    return "hbm"


def issue_side(workload, launch_s):
    """Issue-side figures of the dominant kernel from the rocprofv3 --pmc SQ pass of the same workload
    (profiles/issue.json, written by tools/collect_profiles.sh): vector instructions per launch, the
    fraction of the chip's vector issue slots they fill over THIS run's launch time (1024 SIMDs, a wave64
    vector instruction occupies its SIMD for 4 cycles at 2.4 GHz: an upper bound on the rate), and the
    shares of wave-cycles with an instruction in flight / waiting."""
    tp = os.path.join(REPO, "profiles", "issue.json")
    if not os.path.exists(tp):
        return None
    try:
        rec = json.load(open(tp)).get(workload)
    except Exception:
        return None
    if not rec:
        return None
    out = dict(rec)
    if rec.get("valu_insts_per_launch") and launch_s > 0:
        out["valu_issue_frac"] = rec["valu_insts_per_launch"] * 4.0 / (1024 * launch_s * 2.4e9)
    out["source"] = "profiles/issue.json (rocprofv3 --pmc SQ_* pass, not this run)"
    return out


def dominant_kernel(workload, learning, info):
    """Name of the kernel family the launch average is dominated by (profiles/*_kernel_stats.csv)."""
    if info.get("p2p_fused") and not learning:
        return "k_gibbs_seg_tab_p2p"     # a shard that exchanges its boundary inside its class launches
    if not info["nfast"]:
        return "k_learn_phase" if learning else "k_gibbs_phase"
    if workload.startswith("lr") or workload.startswith("boolw"):
        if learning:
            # entry-parallel groups (+ hubs, rest tiles); graphs with categorical variables: the four-waves-per-SIMD twin
            return "k_learn_ep_w4" if workload.startswith("lr") else "k_learn_ep"
        # categorical graphs whose value array stays in the L2s run the register-capped twin (nsk_gibbs.hip)
        small = info["nvar"] * info["value_bytes"] <= (24 << 20)
        return "k_gibbs_ep_w5" if workload.startswith("lr") and small else "k_gibbs_ep"
    if info["ztab_entries"]:
        # wide quads (one lane samples four consecutive positions): laid out from 400 000 sampled variables per handle on; the
        # learning launches take them from 12 000 quads per class launch on (nsk_internal.h NSK_WIDE_LEARN_MIN_QUADS)
        wide = info.get("wide_quads", 0) * 2 >= max(1, info.get("tab_quads", 0)) and info["value_bytes"] == 1
        if wide and learning and info["wide_quads"] // max(1, info["ncolors"]) >= 12000:
            return "k_learn_seg_tabw"
        if wide and not learning:
            return "k_gibbs_seg_tabw"
        return "k_learn_seg_tab" if learning else "k_gibbs_seg_tab"
    return "k_learn_seg" if learning else "k_gibbs_seg"


def spawn_ranks(args):
    """`python bench.py --gpus N` outside torch.distributed.run: start it as a CHILD process (this
    parent has not touched the GPU, and a process that has must never exec), relay its output and
    exit with its code."""
    import random
    import socket
    # (a port outside the kernel's ephemeral range: one probed with bind(0) can be handed to an outgoing connection
    #  before the rendezvous server binds it)
    port = None
    rng = random.Random(os.getpid() * 7919 + int.from_bytes(os.urandom(4), "little"))
    for _ in range(200):
        cand = rng.randrange(20000, 32000)
        with socket.socket() as s:
            try:
                s.bind(("127.0.0.1", cand))
            except OSError:
                continue
        port = cand
        break
    if port is None:
        sys.exit("bench.py: no free rendezvous port in 20000-31999")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith('{"metric"')]
    if lines:
        print(lines[-1])
    else:
        sys.stdout.write(proc.stdout)
    sys.exit(proc.returncode)


def dry_run(args, dist, rank, world, rows, cols, learning):
    """NSK_BENCH_DRYRUN=1 (CPU test hook): everything of the N-rank launch that needs no GPU -- the
    self-spawn, the rendezvous, the shard formula, the host-side plan of every rank's partition and
    the boundary lists the ranks agree on -- then one JSON line with value 0."""
    import io
    from contextlib import redirect_stdout
    import numbskull_amd
    from numbskull_amd.distributed import shard_range, plan_boundaries, gather_needs
    ns = numbskull_amd.NumbSkull(quiet=True, seed=args.seed, head_by_vid=args.workload.startswith("lr"))
    own = shard_range(rank, world, rows * cols)
    if world > 1:
        g, gids, own_local, _, _, _ = build_shard(rows, cols, learning, args.workload, own[0], own[1])
    else:
        g = build_graph(rows, cols, learning, name=args.workload)
    w, v, f, fm, dm, edges = g
    with redirect_stdout(io.StringIO()):
        if world > 1:
            ns.loadFactorGraph(w, v, f, fm, dm, int(edges), own_range=own_local, global_ids=gids)
        else:
            ns.loadFactorGraph(w, v, f, fm, dm, int(edges))
    fg = ns.factorGraphs[0]
    color, info = fg.plan()
    needs = gids[fg.ghost_needs(host_only=True)].astype(np.int32) if world > 1 else np.zeros(0, np.int32)
    nb = int(len(needs))
    sampled = int((color >= 0).sum())
    rss = peak_rss_gb()
    if world > 1:
        import torch
        t = torch.tensor([sampled, nb], dtype=torch.int64)
        dist.all_reduce(t)
        sampled, nb = int(t[0]), int(t[1])
        t = torch.tensor([rss], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        rss = float(t[0])
        lists, slot = plan_boundaries(gather_needs(dist, torch, needs, world, "cpu"), world, rows * cols)
        nb = int(sum(len(x) for x in lists))
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "variable-updates/sec", "value": 0.0, "unit": "variable-updates/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "dry_run": True,
                          "exchange_ladder": [{"rank": r, "rung": "dry run (no GPU): fused into the table launches -> peer-to-peer "
                                               "exchange kernels -> native RCCL loop -> torch.distributed loop, first that sets up and "
                                               "self-tests on EVERY rank", "tried": [], "seconds": {}} for r in range(world)],
                          "config": {"name": args.workload, "sampled_total": sampled, "boundary_total": nb,
                                     "colors": info["ncolors"], "variables_held_by_rank0": int(len(v)),
                                     "peak_rss_gb_max_over_ranks": round(rss, 2)}}))


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

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)           # does not return

    import torch
    import numbskull_amd
    from numbskull_amd import _lib
    from numbskull_amd.distributed import PartitionedSampler, shard_range

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # test hooks for a box with fewer GPUs than ranks: NSK_BENCH_ONE_DEVICE puts every rank on
        # device 0, NSK_BENCH_BACKEND=gloo replaces RCCL (the library then falls back to the
        # torch.distributed exchange loop); the driver's runs use neither
        if os.environ.get("NSK_BENCH_ONE_DEVICE"):
            local_rank = 0
        if not os.environ.get("NSK_BENCH_DRYRUN"):
            torch.cuda.set_device(local_rank)
        dist.init_process_group(os.environ.get("NSK_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)

    rows, cols, learning = WORKLOADS[args.workload]
    is_grid = args.workload.startswith("ising")
    if os.environ.get("NSK_BENCH_DRYRUN"):
        return dry_run(args, dist, rank, world, rows, cols, learning)
    nvar = rows * cols
    ns = numbskull_amd.NumbSkull(quiet=True, device=local_rank, seed=args.seed,
                                 head_by_vid=args.workload.startswith("lr"))
    own = shard_range(rank, world, nvar)
    import io
    from contextlib import redirect_stdout
    if world > 1:
        # every rank holds only its shard -- owned variables, the ghosts they read, the factors that
        # touch them (the reference's minions load their partition only, salt/src/numbskull_minion.py:185);
        # the LR workloads never materialise the whole graph (graphgen.mixed_lr_shard)
        g, gids, own_local, nfactor_total, nweight_total, t_gen = build_shard(rows, cols, learning, args.workload, own[0], own[1])
        w, v, f, fm, dm, edges = g
        if nfactor_total is None:                   # (factors with members in several shards are held by each of them)
            nfactor_total = -1
        t_load = time.time()
        with redirect_stdout(io.StringIO()):
            ns.loadFactorGraph(w, v, f, fm, dm, int(edges), own_range=own_local, global_ids=gids)
    else:
        t_gen = time.time()
        g = build_graph(rows, cols, learning, name=args.workload)
        t_gen = time.time() - t_gen
        nfactor_total, nweight_total = len(g[2]), len(g[0])
        w, v, f, fm, dm, edges = g
        t_load = time.time()
        with redirect_stdout(io.StringIO()):
            ns.loadFactorGraph(w, v, f, fm, dm, int(edges))
    fg = ns.factorGraphs[0]
    L, h = _lib.lib(), fg._engine()
    t_load = time.time() - t_load
    info = fg.info()
    if world > 1:
        # the library is pointed at torch's current stream: a stream of its own, so that the sweep
        # sequences can be captured into hipGraphs (the legacy default stream cannot be captured)
        torch.cuda.set_stream(torch.cuda.Stream())
    sampler = PartitionedSampler(fg, dist, torch, rank, world, nvar_global=nvar) if world > 1 else None
    lr = (1e-7, 0.95, 2, 0.01, 1)       # step, decay, L2, reg_param, truncation (config #3)
    if args.workload.startswith("lr") or args.workload.startswith("boolw"):
        lr = (1e-3, 0.95, 2, 0.01, 1)

    def run(n, burnin=False):
        if learning:
            if world > 1:
                sampler.learn(n, *lr)
            else:
                _lib.check(L.nsk_learn_sweeps(h, n, lr[0], 1.0, lr[2], lr[3], lr[4], 0))
        else:
            if world > 1:
                sampler.gibbs(n, True, burnin)
            else:
                _lib.check(L.nsk_gibbs_sweeps(h, n, 1, int(burnin)))

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # warm-up sweeps are burn-in sweeps (factorgraph.py:129-143: same kernels, no tally), so the
    # tallies the state checks read cover the timed sweeps only, not the transient from the
    # all-zero initial state
    # (the statistical state checks below need a chain that has left the transient from the all-zero state:
    # at least 10 untimed burn-in sweeps whatever --warmup is; the extra ones run before the warm-up proper)
    burn_extra = max(0, 10 - args.warmup)
    if burn_extra:
        run(burn_extra, True)
    run(args.warmup, True)
    import ctypes as C
    blocks = timed_blocks(run, fence, L, h, args.steps, REPEATS)
    if world > 1:        # every block: the slowest rank's time
        t = torch.tensor([b[0] for b in blocks], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        blocks = [(float(t[i].item()), b[1], b[2]) for i, b in enumerate(blocks)]
    dt, ms_ev_v, launches_v = median_block(blocks)
    if sampler is not None:
        sampler.check()                  # a peer-to-peer exchange that timed out fails the run here

    class _V(object):       # (keeps the names the report below uses)
        def __init__(self, v):
            self.value = v
    ms_ev, launches = _V(ms_ev_v), _V(launches_v)

    # invariants of the state the timed sweeps left behind (every rank checks its own shard)
    checks = state_checks(fg, info, args.steps * REPEATS, (rows, cols) if is_grid else None, learning)
    # hard invariants decide the exit code; the statistical ones (mean marginal, agreement with the
    # CPU chain) are always reported but only count once the chain has been burnt in (>= 10 warm-up
    # sweeps): a 2-sweep warm-up of a profiling pass still sits in the transient from the all-zero
    # state -- the line then says parity.statistics = "skipped (warm-up < 10 sweeps)" instead of passing
    # them silently
    burnt_in = args.warmup + burn_extra >= 10
    ok_local = all(bool(x) for k, x in checks.items()
                   if k.endswith("_in_bounds") or k in ("values_in_domain", "weights_finite")
                   or (burnt_in and k.endswith("_ok")))
    checks["statistics_count"] = burnt_in
    checks["statistics"] = "counted" if burnt_in else "skipped (warm-up < 10 sweeps)"
    checks["burn_in_sweeps"] = args.warmup + burn_extra      # untimed sweeps before the first timed block
    if world > 1:
        t = torch.tensor([1.0 if ok_local else 0.0], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        checks["all_ranks_ok"] = bool(t.item() > 0.5)
        ok_local = checks["all_ranks_ok"]

    phases = None
    if world > 1 and (not learning or sampler.p2p):      # per-phase timings of one shard's sweep (diagnostic)
        phases = sampler.phase_timings(20, learn=(lr[0], lr[2], lr[3], lr[4]) if learning else None)
        sampler.check()
        phases["exchange_path"] = "peer-to-peer writes + flags" if sampler.p2p else (
            "native RCCL all-gather" if sampler.native else "torch.distributed all-gather")
    # N ranks on the metric config: a second line that is not bound by launch latency -- the 100M grid (12.5M cells per
    # rank at N = 8; a 1.25M-cell shard of the 10M grid cannot scale below one launch floor per class)
    multi_also = None
    if world > 1 and args.workload == "ising10m" and not args.no_extra:
        multi_also = multi_side_run("ising100m", args, dist, torch, rank, world, local_rank)
    ladder = None
    if world > 1:                       # every rank's rung of the exchange ladder (numbskull_amd/distributed.py)
        ladder = [None] * world
        dist.all_gather_object(ladder, sampler.ladder)
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
        traffic, traffic_src = None, None
        tp = os.path.join(REPO, "profiles", "traffic.json")
        if os.path.exists(tp) and world == 1:      # PMC bytes were collected for the one-GPU launch
            try:
                tj = json.load(open(tp))
                traffic = tj.get(args.workload)
                traffic_src = "profiles/traffic.json (%s)" % tj.get("_source", "rocprofv3 --pmc passes, not this run")
            except Exception:
                traffic = None
        out = {
            "metric": "variable-updates/sec", "value": nvar * args.steps / dt,
            "unit": "variable-updates/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt * 1e3 / args.steps,
            "repeats": spread(blocks, args.steps),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            # potentials, exp and the draw are float64 (bit-equal to the float64 oracle); the table
            # kernels tabulate that float64 decision per neighbourhood and compare 53-bit integers
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": ("%d boolean variables, %d ISTRUE/OR/EQUAL factors with one weight each, "
                                    "inference only, chromatic scan, seed %d" % (nvar, nfactor_total, args.seed))
                       if args.workload.startswith("boolw4m") else
                       ("mixed-arity LR graph: %d variables (25%% categorical), %d factors, %d weights, %s; head_by_vid (SURVEY.md "
                        "section 8d: the reference's literal head index, inference.py:243, is out of bounds on this graph)"
                        % (nvar, nfactor_total, nweight_total, "learning" if learning else "inference"))
                       if args.workload.startswith("lr") else
                       "%dx%d Ising grid (%d binary variables, %d EQUAL factors), %s, "
                       "chromatic scan, seed %d"
                       % (rows, cols, nvar, nfactor_total, "learning (2 free weights, L2)"
                          if learning else "inference only, weight 0.1 fixed", args.seed),
                       "name": args.workload, "partition": "range by variable id, %d shard(s)%s" % (
                           world, ", shard-local graphs (%d of %d variables held by rank 0)" % (len(v), nvar) if world > 1 else ""),
                       "colors": info["ncolors"], "value_bytes": info["value_bytes"],
                       "generate_s": round(t_gen, 2), "load_and_compile_s": round(t_load, 2),
                       "compile_s": round(info["compile_seconds"], 2),
                       "device_bytes": info["device_bytes"]},
            # bound: the roofline the fraction is taken against (byte / integer work: HBM; there is no MFMA on this path);
            # limited_by: what the profiles say keeps the kernel below it
            "roofline": {"bound": "hbm", "limited_by": roofline_bound(dominant_kernel(args.workload, learning, dict(info, p2p_fused=fg.info()["p2p_fused"]))),
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         # launch time / the time the launch's bytes (PMC traffic when collected, else the
                         # layout bytes) would take at the HBM peak: 1 = at the memory floor
                         "time_over_memory_floor": launch_s / ((traffic or lay_per_launch) / (HBM_PEAK_GBS * 1e9)),
                         "issue": issue_side(args.workload, launch_s),
                         "traffic_source": traffic_src,
                         "traffic_GBs": (traffic / launch_s / 1e9) if traffic else None,
                         "layout_bytes_per_update": lay_sweep * world / nvar,
                         "sweep_stream_bytes": lay_sweep,
                         "stream_fits_infinity_cache": bool(lay_sweep < INFINITY_CACHE_BYTES),
                         "alg_bytes_per_update_csr": alg_sweep * world / nvar,
                         "csr_model_GBs": alg_sweep * args.steps / (ms_ev.value / 1e3) / 1e9,
                         "kernel": dominant_kernel(args.workload, learning, dict(info, p2p_fused=fg.info()["p2p_fused"])),
                         "stream_copy_GBs": copy_gbs,
                         "frac_of_stream_copy": (achieved / copy_gbs) if copy_gbs else None,
                         "launches": nlaunch, "avg_launch_us": launch_s * 1e6},
            "parity": checks,
        }
        if phases is not None:
            out["phases_us"] = phases      # rank 0's shard: sweep kernels / exchange, each with its launch latency
        if ladder is not None:
            out["exchange_ladder"] = ladder
        if learning:
            out["learn_clipped"] = fg.info()["learn_clipped"]
            out["learn_hyper"] = {"step": lr[0], "regularization": lr[2], "reg_param": lr[3],
                                  "learn_cap": info["learn_cap"]}
        if not args.no_cpu_baseline and world == 1:
            cb = cpu_baseline(fg, learning, grid=(rows, cols) if is_grid else None,
                              head_by_vid=args.workload.startswith("lr"), lr=lr)
            agree, mm = cb.pop("_agreement"), cb.pop("_mean_marginal")
            cb.pop("_sweeps")
            out["cpu_baseline"] = cb
            if agree is not None and "edge_agreement" in checks:
                # L3 statistic: both samplers target the same distribution,
                # so the agreement fraction over the 2*10^7 grid edges must coincide (the standard
                # error of one configuration's fraction is ~1e-4)
                checks["edge_agreement_cpu"] = agree
                checks["edge_agreement_diff"] = abs(agree - checks["edge_agreement"])
                checks["edge_agreement_ok"] = bool(checks["edge_agreement_diff"] < 1e-3)
                checks["mean_marginal_cpu"] = mm
                ok_local = ok_local and (checks["edge_agreement_ok"] or not burnt_in)
        if world == 1 and args.workload == "ising10m" and not args.no_extra:
            out["also"] = {"ising1m": side_run("ising1m", args.seed, 400, 100),
                           "ising10m_learn": side_run("ising10m_learn", args.seed, 100, 20),
                           "ising40m": side_run("ising40m", args.seed, 50, 10),
                           # BASELINE configs[4] on one GPU: the LR graph at a tenth of its size (both sweeps) and its
                           # learning sweep at full size
                           "lr5m": side_run("lr5m", args.seed, 20, 5),
                           "lr5m_learn": side_run("lr5m_learn", args.seed, 10, 3),
                           "lr50m_learn": side_run("lr50m_learn", args.seed, 5, 2)}
            out["roofline"]["frac_is"] = ("bytes the compiled layout moves per launch (layout_bytes_per_update x the launch's "
                                          "updates) / launch time / HBM peak.  With implicit adjacency and wide quads the table "
                                          "kernel reads one dword per member slot for four positions and no per-lane stream: it "
                                          "sits on a chain of dependent round trips (limited_by = 'issue'; roofline.issue, "
                                          "time_over_memory_floor), not on HBM; csr_model_GBs prices the same sweep in SURVEY "
                                          "8(d)'s CSR layout.  Both grids' sweeps fit the Infinity Cache "
                                          "(stream_fits_infinity_cache); the 100M grid (--workload ising100m) is the one beyond it")
        if multi_also is not None:
            out["also"] = {"ising100m": multi_also}
        checks["ok"] = bool(ok_local)
        # the counter figures above (roofline.traffic / .issue, also.*.traffic_per_launch) are read from profiles/, not
        # measured by this run: say so at top level when they were collected at another library
        out["traffic_stale"] = bool(profiles_stale())
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    if rank == 0 and not ok_local:
        sys.exit(3)                  # a fast sweep with a broken state is not a result


if __name__ == "__main__":
    main()
