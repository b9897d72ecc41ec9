"""CPU-side checks: the C-ABI library loads and exports what include/numbskull_amd.h declares,
the host entry points (index build, parsers, graph validation) behave like the reference, and the
product never reaches into oracle/."""

import ctypes as C
import os
import re
import sys

import numpy as np
import pytest

from conftest import graph_from, GOLDEN, REPO
import numbskull_amd
from numbskull_amd import _lib, dataloading, graphgen
from numbskull_amd.numbskulltypes import Weight, Variable, Factor, FactorToVar, VarToFactor
from util import quiet, session


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(REPO, "include", "numbskull_amd.h")).read()
    declared = set(re.findall(r"\b(nsk_[a-z_0-9]+)\s*\(", header))
    declared -= {"nsk_graph_desc", "nsk_graph_info"}
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    L = C.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name
    assert b"gfx950" in _lib.lib().nsk_version()


def test_record_layouts_match_header():
    header = open(os.path.join(REPO, "include", "numbskull_amd.h")).read()
    for name, dt in (("nsk_weight", Weight), ("nsk_variable", Variable), ("nsk_factor", Factor),
                     ("nsk_ftv", FactorToVar), ("nsk_vtf", VarToFactor)):
        m = re.search(r"\}\s*%s;\s*/\*\s*(\d+) B" % name, header)
        assert m and int(m.group(1)) == dt.itemsize, name


def test_product_does_not_touch_oracle():
    for root, _, files in os.walk(os.path.join(REPO, "numbskull_amd")):
        for fn in files:
            if fn.endswith((".py", ".h", ".hip", ".cpp", ".c")) or fn == "Makefile":
                text = open(os.path.join(root, fn)).read()
                for line in text.splitlines():
                    if re.search(r"^\s*(from|import)\s+oracle|#include.*oracle|libnsk_oracle", line):
                        raise AssertionError("%s reaches into oracle/: %s" % (fn, line))


@pytest.mark.parametrize("tag", ["grid4x5", "mixed", "skiplast", "pairs", "lf"])
def test_compute_var_map_matches_reference(golden, tag):
    z = golden("g2_index_build.npz")
    w, v, f, fm, dm, _ = graph_from(z, tag)
    skip = z[tag + "_in_factors_to_skip"] if tag + "_in_factors_to_skip" in z.files \
        else np.empty(0, np.int64)
    v = v.copy()
    nedges = int(f["arity"].sum() - f["arity"][skip].sum())
    vmap, fi = dataloading.new_index(v, nedges)
    dataloading.compute_var_map(v, f, fm, vmap, fi, dm, skip)
    assert np.array_equal(v, z[tag + "_out_variable"])
    assert np.array_equal(vmap, z[tag + "_out_vmap"])
    assert np.array_equal(fi, z[tag + "_out_factor_index"])


def test_compute_var_map_index_error_like_reference():
    # skipping a factor whose members are not in the last slot overflows factor_index in the
    # reference (IndexError, dataloading.py:64)
    w, v, f, fm, dm, _ = graphgen.ising_grid(3, 3)
    skip = np.array([0], np.int64)
    vmap, fi = dataloading.new_index(v, int(f["arity"].sum()) - 2)
    with pytest.raises(IndexError):
        dataloading.compute_var_map(v, f, fm, vmap, fi, dm, skip)


def test_compute_var_map_does_not_depend_on_the_thread_count():
    """The edges, factors and slots of the index build go over the host threads (atomic counts and cursors, every slot
    sorted afterwards): 1 thread and 8 threads of a subprocess each build the same arrays as this process."""
    import hashlib
    import subprocess
    code = ("import sys, hashlib, numpy as np; sys.path.insert(0, %r)\n"
            "from numbskull_amd import dataloading, graphgen\n"
            "w, v, f, fm, dm, e = graphgen.mixed_lr_graph(30000, seed=3, nweights=50)\n"
            "skip = np.array([5, 17, 17000], np.int64)\n"
            "vmap, fi = dataloading.new_index(v, int(f['arity'].sum()))\n"
            "dataloading.compute_var_map(v, f, fm, vmap, fi, dm, skip)\n"
            "print(hashlib.md5(vmap.tobytes() + fi.tobytes()).hexdigest())\n") % REPO
    out = set()
    for threads in ("1", "8"):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, NSK_COMPILE_THREADS=threads))
        assert r.returncode == 0, r.stderr[-2000:]
        out.add(r.stdout.strip())
    assert len(out) == 1, out


def test_state_layout_matches_the_reference_formulas():
    """nsk_state_layout against factorgraph.py:41-53 written with numpy: cstart, initial values, Z's and fids' widths."""
    w, v, f, fm, dm, e = graphgen.mixed_lr_graph(20000, seed=11, nweights=40)
    ns = numbskull_amd.NumbSkull(quiet=True)
    ns.loadFactorGraph(w, v, f, fm, dm, e)
    fg = ns.factorGraphs[0]
    card = fg.variable["cardinality"].astype(np.int64)
    cstart = np.zeros(len(card) + 1, np.int64)
    np.cumsum(np.where(card == 2, 1, card), out=cstart[1:])
    assert np.array_equal(fg.cstart, cstart)
    assert np.array_equal(fg.var_value[0], fg.variable["initialValue"]) and np.array_equal(fg.var_value_evid[0], fg.variable["initialValue"])
    assert fg.Z.shape[1] == int(card.max()) and fg.fids.shape[1] == 2 * int(fg.vmap["factor_index_length"].max())
    assert len(fg.count) == int(cstart[-1]) == len(fg.marginals)


@pytest.mark.parametrize("name", ["coin", "domains"])
def test_file_loader_matches_reference(golden, name):
    z = golden("g2_index_build.npz")
    d = os.path.join(GOLDEN, "test_coin" if name == "coin" else "domains_graph")
    ns = numbskull_amd.NumbSkull(directory=d, quiet=True)
    quiet(ns.loadFGFromFile)
    fg = ns.factorGraphs[0]
    for attr in ("weight", "variable", "factor", "fmap", "vmap", "factor_index", "cstart"):
        assert np.array_equal(getattr(fg, attr), z["%s_out_%s" % (name, attr)]), attr
    assert np.array_equal(fg.var_value[0], fg.variable["initialValue"])
    assert fg.weight_value.shape == (1, len(fg.weight))


def test_meta_with_trailing_path_fields(tmp_path):
    """The reference's own test/graph.meta carries 4 extra path fields."""
    src = os.path.join(GOLDEN, "test_coin")
    for fn in ("graph.weights", "graph.variables", "graph.factors"):
        (tmp_path / fn).write_bytes(open(os.path.join(src, fn), "rb").read())
    (tmp_path / "graph.meta").write_text(open(os.path.join(src, "graph.meta.orig")).read())
    ns = numbskull_amd.NumbSkull(directory=str(tmp_path), quiet=True)
    quiet(ns.loadFGFromFile)
    assert len(ns.factorGraphs[0].variable) == 18


def test_writer_roundtrip(tmp_path):
    g = graphgen.ising_pairs(7, seed=1)
    graphgen.write_graph(str(tmp_path), g[0], g[1], g[2], g[3])
    ns = numbskull_amd.NumbSkull(directory=str(tmp_path), quiet=True)
    quiet(ns.loadFGFromFile)
    fg = ns.factorGraphs[0]
    assert np.array_equal(fg.weight, g[0])
    for k in ("isEvidence", "initialValue", "dataType", "cardinality"):
        assert np.array_equal(fg.variable[k], g[1][k])
    assert np.array_equal(fg.factor, g[2])
    assert np.array_equal(fg.fmap, g[3])


def test_reads_reference_generator_output(tmp_path):
    """oracle/_ref/ising is the reference's own ising/ising.cpp compiled where it lies; our loader
    must read what it writes (skipped where the reference tree never existed)."""
    exe = os.path.join(REPO, "oracle", "_ref", "ising")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ising not built")
    import subprocess
    subprocess.check_call([exe], cwd=str(tmp_path), stdout=subprocess.DEVNULL)
    ns = numbskull_amd.NumbSkull(directory=str(tmp_path), quiet=True)
    quiet(ns.loadFGFromFile)
    fg = ns.factorGraphs[0]
    assert len(fg.variable) == 2000 and len(fg.factor) == 3000 and len(fg.weight) == 3
    assert set(np.unique(fg.factor["factorFunction"])) == {3, 4}
    assert np.all(fg.variable["isEvidence"] == 1)
    # and our writer emits the same bytes for the same graph
    out = tmp_path / "again"
    graphgen.write_graph(str(out), fg.weight, fg.variable, fg.factor, fg.fmap)
    for fn in ("graph.weights", "graph.variables", "graph.factors", "graph.meta"):
        assert (out / fn).read_bytes() == (tmp_path / fn).read_bytes(), fn


def test_unknown_factor_function_raises_not_implemented():
    """inference.py:410-413 raises NotImplementedError; validation happens at graph creation,
    before any device is touched, so this runs without a GPU."""
    g = list(graphgen.ising_grid(3, 3))
    g[2] = g[2].copy()
    g[2]["factorFunction"][2] = 5
    ns, fg = session(tuple(g))
    with pytest.raises(NotImplementedError):
        fg.inference(0, 1)


def test_literal_head_index_out_of_range_raises_index_error():
    g = graphgen.mixed_lr_graph(300, seed=1)
    ns, fg = session(g)
    with pytest.raises(IndexError):
        fg.inference(0, 1)


def test_fails_loudly_without_gpu():
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    ns, fg = session(graphgen.ising_grid(4, 4))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        fg.inference(0, 1)


def test_factor_table_and_api_surface():
    from numbskull_amd import inference, numbskull
    assert inference.FACTORS["IMPLY_MLN"] == 13 and inference.FUNC_UFO == 30
    assert len(inference.FACTORS) == 25
    ns = numbskull_amd.NumbSkull(n_inference_epoch=7, quiet=True)
    assert ns.n_inference_epoch == 7 and ns.sample_evidence is True and ns.nthreads == 1
    assert ns.stepsize == 0.01 and ns.decay == 0.95 and ns.regularization == 2
    assert [o["dest"] for _, o in numbskull.flags] == ["sample_evidence", "learn_non_evidence",
                                                       "quiet", "verbose"]
    assert numbskull_amd.__version__.startswith("0.1.1")


def _graph_from_spec(nvar, spec, card=None, weights=(0.5,)):
    """tiny graph from [(function, [member ids])]"""
    variable = np.zeros(nvar, Variable)
    variable["cardinality"] = 2 if card is None else card
    factor = np.zeros(len(spec), Factor)
    fmap = np.zeros(sum(len(m) for _, m in spec), FactorToVar)
    e = 0
    for i, (fn, members) in enumerate(spec):
        factor[i] = (fn, i % len(weights), 1.0, len(members), e)
        for m in members:
            fmap[e]["vid"] = m
            e += 1
    weight = np.zeros(len(weights), Weight)
    weight["initialValue"] = weights
    return weight, variable, factor, fmap, np.zeros(nvar, np.bool_), e


def test_plan_edge_cases():
    """empty graph, isolated variables, cardinality-1 variables, a factor over one variable twice,
    a hub factor that forces many colours -- planning must succeed and colour validly"""
    from util import check_coloring
    ns, fg = session(_graph_from_spec(0, []))
    color, info = fg.plan()
    assert info["nowned"] == 0 and info["ncolors"] == 0 and len(color) == 0
    ns, fg = session(_graph_from_spec(5, []))                       # no factors at all
    color, info = fg.plan()
    assert info["ncolors"] == 1 and info["nfast"] == 5
    ns, fg = session(_graph_from_spec(3, [(4, [0]), (3, [1, 1]), (1, [2, 2, 0])],
                                      card=np.array([1, 2, 2])))
    color, info = fg.plan()
    check_coloring(fg, color)
    ns, fg = session(_graph_from_spec(40, [(2, list(range(40)))]))  # one AND over 40 variables
    color, info = fg.plan()
    assert info["ncolors"] == 40 and info["ngeneric"] == 40         # > 6 other members: generic path
    check_coloring(fg, color)


def test_bad_graphs_are_rejected_with_reference_like_errors():
    g = list(_graph_from_spec(3, [(3, [0, 1])]))
    g[3] = g[3].copy()
    g[3]["vid"][1] = 7                                               # member outside the variables
    with pytest.raises(IndexError):
        session(tuple(g))                                            # compute_var_map faults first
    g = list(_graph_from_spec(3, [(3, [0, 1])]))
    g[2] = g[2].copy()
    g[2]["weightId"][0] = 5
    ns, fg = session(tuple(g))
    with pytest.raises(IndexError):
        fg.plan()
    g = list(_graph_from_spec(2, [(21, [0])]))                       # DP_GEN_LF_ACCURACY reads 2 members
    ns, fg = session(tuple(g))
    with pytest.raises(IndexError):
        fg.plan()


def test_evidence_value_outside_domain_is_rejected():
    """An evidence value of a dataType-1 variable selects its factor list (learning.py:61-62 with
    get_factor_id_range): outside [0, cardinality) the reference reads another variable's lists;
    the graph compiler refuses (IndexError)."""
    g = list(_graph_from_spec(3, [(14, [0, 1])], card=np.array([3, 3, 2])))
    g[1] = g[1].copy()
    g[1]["dataType"][:2] = 1
    ns, fg = session(tuple(g))
    fg.plan()                                                        # fine as it stands
    g[1]["isEvidence"][0] = 1
    g[1]["initialValue"][0] = 5
    ns, fg = session(tuple(g))
    with pytest.raises(IndexError):
        fg.plan()


def test_empty_shard_is_empty_and_unflagged_zero_range_is_the_whole_graph():
    """shard_range(0, world, nvar) is (0, 0) when nvar < world: with own_range given that is an EMPTY
    shard (NSK_FLAG_PARTITION), not the whole graph."""
    from numbskull_amd import graphgen
    from numbskull_amd.distributed import shard_range
    g = graphgen.ising_grid(1, 3, weight=0.2)
    assert shard_range(0, 8, 3) == (0, 0)
    ns = numbskull_amd.NumbSkull(quiet=True)
    ns.loadFactorGraph(*[x.copy() if isinstance(x, np.ndarray) else x for x in g[:5]], int(g[5]),
                       own_range=shard_range(0, 8, 3))
    color, info = ns.factorGraphs[0].plan()
    assert info["nowned"] == 0 and (color < 0).all()
    ns, fg = session(g)
    color, info = fg.plan()
    assert info["nowned"] == 3
    owned = 0
    for r in range(8):                                               # every variable has exactly one owner
        ns = numbskull_amd.NumbSkull(quiet=True)
        ns.loadFactorGraph(*[x.copy() if isinstance(x, np.ndarray) else x for x in g[:5]], int(g[5]),
                           own_range=shard_range(r, 8, 3))
        owned += ns.factorGraphs[0].plan()[1]["nowned"]
    assert owned == 3


def test_write_probabilities_matches_the_reference_format(tmp_path):
    """nsk_write_probabilities == FactorGraph.dump_probabilities (factorgraph.py:216-229): binary
    variables one line with value 1, others one line per domain value taken from vmap.value."""
    g = _graph_from_spec(4, [(14, [0, 1]), (4, [2]), (3, [2, 3])], card=np.array([3, 4, 2, 2]))
    g = list(g)
    g[1] = g[1].copy()
    g[1]["dataType"][:2] = 1
    ns, fg = session(tuple(g))
    rng = np.random.default_rng(1)
    fg.count[:] = rng.integers(0, 7, len(fg.count))
    fg.vmap["value"][:] = np.arange(len(fg.vmap))[::-1] * 3
    out = tmp_path / "p.txt"
    fg.dump_probabilities(str(out), 7)
    want = []
    for i, v in enumerate(fg.variable):
        if v["cardinality"] == 2:
            want.append('%d %d %.3f\n' % (i, 1, float(fg.count[fg.cstart[i]]) / 7))
        else:
            for k in range(v["cardinality"]):
                want.append('%d %d %.3f\n' % (i, fg.vmap[v["vtf_offset"] + k]["value"],
                                              float(fg.count[fg.cstart[i] + k]) / 7))
    assert out.read_text() == "".join(want)


def test_write_probabilities_refuses_a_malformed_variable_array(tmp_path):
    """A vtf_offset / tally slot beyond the arrays handed in is NSK_E_INDEX (IndexError), checked
    before anything is written -- the reference would raise IndexError from numpy."""
    g = _graph_from_spec(3, [(14, [0, 1]), (4, [2])], card=np.array([3, 2, 2]))
    g = list(g)
    g[1] = g[1].copy()
    g[1]["dataType"][:1] = 1
    ns, fg = session(tuple(g))
    out = tmp_path / "p.txt"
    fg.variable["vtf_offset"][0] = len(fg.vmap) - 1            # 3 domain values do not fit behind it
    with pytest.raises(IndexError):
        fg.dump_probabilities(str(out), 1)
    assert not out.exists()
    fg.variable["vtf_offset"][0] = 0
    fg.cstart[2] = len(fg.count) + 5
    with pytest.raises(IndexError):
        fg.dump_probabilities(str(out), 1)


def test_diagnostic_switches_need_nsk_diag(monkeypatch):
    """Layout switches (NSK_NO_FAST & co.) change which kernels a graph compiles to -- and therefore its
    sample stream -- so an inherited environment variable alone must not do that: they are read only
    together with NSK_DIAG=1."""
    from numbskull_amd import graphgen
    g = graphgen.ising_grid(16, 16, weight=0.2)
    monkeypatch.delenv("NSK_DIAG", raising=False)
    monkeypatch.delenv("NSK_NO_FAST", raising=False)
    base = session(g)[1].plan()[1]
    assert base["nfast"] == 256
    monkeypatch.setenv("NSK_NO_FAST", "1")
    assert session(g)[1].plan()[1]["nfast"] == 256          # ignored without NSK_DIAG
    monkeypatch.setenv("NSK_DIAG", "1")
    assert session(g)[1].plan()[1]["nfast"] == 0            # honoured with it


_PLAN_SCRIPT = r"""
import hashlib, json, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from tests.util import session
from numbskull_amd import graphgen
out = {}
for name, g, kw in (("boolw", graphgen.boolean_weighted_graph(60000, seed=9), {}),
                    ("lr", graphgen.mixed_lr_graph(60000, seed=4, nweights=700), {"head_by_vid": True}),
                    ("grid", graphgen.ising_grid(150, 220, weight=0.1), {})):
    if name == "boolw":
        g[0]["isFixed"] = False
    color, info = session(g, **kw)[1].plan()
    out[name] = [hashlib.sha256(np.ascontiguousarray(color).tobytes()).hexdigest(), info]
print(json.dumps(out, sort_keys=True))
"""


def test_graph_compiler_is_independent_of_the_thread_count():
    """compile_graph runs its stages over host threads (static index blocks, one thread per colour for the
    class maps, class-by-class recolouring): colours, the path split, every byte count of the plan and the hash
    of every array of the compiled layout (nsk_graph_info.layout_hash) must not depend on how many there are.  (The thread count is read once per process: two subprocesses.)"""
    import json
    import subprocess
    outs = []
    for threads in ("1", "7"):
        env = dict(os.environ, NSK_COMPILE_THREADS=threads, NSK_LAYOUT_HASH="1")
        env.pop("NSK_DIAG", None)
        r = subprocess.run([sys.executable, "-c", _PLAN_SCRIPT, REPO], env=env, capture_output=True, text=True,
                           timeout=600, cwd=REPO)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert outs[0] == outs[1]
    assert outs[0]["boolw"][1]["direct_weights"] > 0 and outs[0]["lr"][1]["ncolors"] >= 2
    assert all(outs[0][k][1]["layout_hash"] != 0 for k in outs[0])        # every array of the layout, hashed
    assert len({outs[0][k][1]["layout_hash"] for k in outs[0]}) == 3


def test_weight_slots_only_on_whole_graph_handles_with_single_factor_weights():
    """Single-factor weights get slots in layout order (nsk_graph_info.weight_slots) on a handle that owns the
    whole graph; a handle that samples a range keeps the caller's numbering (the ranks of a distributed run add
    their weight tables element by element), and so does a graph whose weights are shared."""
    bw = list(graphgen.boolean_weighted_graph(20000, seed=2))
    bw[0]["isFixed"] = False
    whole = session(tuple(bw))[1].plan()[1]
    assert whole["direct_weights"] > 0 and whole["weight_slots"] == 1
    ns = numbskull_amd.NumbSkull(quiet=True)
    ns.loadFactorGraph(*[x.copy() if isinstance(x, np.ndarray) else x for x in bw[:5]], int(bw[5]), own_range=(0, 10000))
    part = ns.factorGraphs[0].plan()[1]
    assert part["direct_weights"] > 0 and part["weight_slots"] == 0
    lr = session(graphgen.mixed_lr_graph(20000, seed=3, nweights=500), head_by_vid=True)[1].plan()[1]
    assert lr["direct_weights"] == 0 and lr["weight_slots"] == 0


def test_a_factor_listed_twice_by_one_variable_is_not_updated_in_place():
    """compute_var_map never lists a factor twice for one (variable, value) (dataloading.py:68-81), but a caller that
    hands prebuilt lists to loadFactorGraphRaw may: such a factor's weight is visited twice by one variable in one
    colour class, so the compiler must keep it on the accumulators (nsk_graph_info.direct_weights)."""
    bw = list(graphgen.boolean_weighted_graph(4000, seed=5))
    bw[0]["isFixed"] = False
    fg = session(tuple(bw))[1]
    base = fg.plan()[1]["direct_weights"]
    assert base > 0
    # variable v's list with its first factor twice: append a second copy of the whole index with the duplicate
    vm, fi = fg.vmap.copy(), fg.factor_index.copy()
    v = int(np.argmax(vm["factor_index_length"] >= 1))
    off, ln = int(vm["factor_index_offset"][v]), int(vm["factor_index_length"][v])
    dup = np.concatenate([fi[off:off + ln], fi[off:off + 1]])
    vm["factor_index_offset"][v] = len(fi)
    vm["factor_index_length"][v] = ln + 1
    fi2 = np.concatenate([fi, dup])
    ns = numbskull_amd.NumbSkull(quiet=True)
    ns.loadFactorGraphRaw(fg.weight, fg.variable, fg.factor, fg.fmap, vm, fi2)
    again = ns.factorGraphs[0].plan()[1]["direct_weights"]
    assert again == base - 1
