/*
 * numbskull_amd.h -- C-ABI of the MI355X-native Gibbs-sweep engine (libnumbskull_amd.so).
 *
 * The reference (HazyResearch/numbskull) has no FFI layer: its seam is the three
 * run_pool(...) call sites in numbskull/factorgraph.py
 *     :141  burn-in   -> inference.gibbsthread(..., burnin=True)
 *     :163  inference -> inference.gibbsthread(..., burnin=False)
 *     :202  learning  -> learning.learnthread(...)
 * whose callee signatures are inference.py:10-13 and learning.py:12-16.  A maintainer makes the
 * reference GPU-backed by replacing those three calls with nsk_gibbs_sweeps / nsk_learn_sweeps on
 * a handle created from the very arrays FactorGraph already owns (INTEGRATION.md shows the ctypes
 * stub).  Everything crosses this boundary as plain pointers and sizes; the record layouts are
 * the reference's packed numpy dtypes (numbskulltypes.py:11-39), so record arrays are passed by
 * pointer without conversion.
 *
 * Conventions: every function returns NSK_OK (0) or a negative NSK_E_* code; nsk_last_error()
 * returns a thread-local human-readable message for the last failure.  Host pointers stay owned
 * by the caller; the library owns device memory.  One in-flight call per handle.
 */
#ifndef NUMBSKULL_AMD_H
#define NUMBSKULL_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NSK_OK 0
#define NSK_E_INVALID (-1)      /* bad argument / inconsistent sizes                              */
#define NSK_E_FACTOR_FUNC (-2)  /* unknown factorFunction: reference raises NotImplementedError
                                   (inference.py:410-413)                                         */
#define NSK_E_INDEX (-3)        /* an index the reference would fault on (IndexError)             */
#define NSK_E_DEVICE (-4)       /* HIP runtime failure or no usable device                        */
#define NSK_E_RANGE (-5)        /* a value does not fit the device representation                 */
#define NSK_E_NOMEM (-6)

/* ---- record layouts = numbskull/numbskulltypes.py:11-39 (packed) ---- */
#pragma pack(push, 1)
typedef struct { uint8_t isFixed; double initialValue; } nsk_weight;                  /*  9 B */
typedef struct { int8_t isEvidence; int64_t initialValue; int16_t dataType;
                 int64_t cardinality; int64_t vtf_offset; } nsk_variable;             /* 27 B */
typedef struct { int16_t factorFunction; int64_t weightId; double featureValue;
                 int64_t arity; int64_t ftv_offset; } nsk_factor;                     /* 34 B */
typedef struct { int64_t vid; int64_t dense_equal_to; } nsk_ftv;                      /* 16 B */
typedef struct { int64_t value; int64_t factor_index_offset;
                 int64_t factor_index_length; } nsk_vtf;                              /* 24 B */
#pragma pack(pop)

/* nsk_graph_desc.flags */
#define NSK_FLAG_HEAD_BY_VID 1  /* IMPLY_MLN / IMPLY_NATURAL_CAT / IMPLY_MLN_CAT read their head as
                                   var_value[fmap[l].vid] instead of the reference's literal
                                   var_value[l] (inference.py:243,277,292)                        */

#define NSK_FLAG_PARTITION 2    /* [own_begin, own_end) is this handle's shard even when it is empty;
                                   without the flag own_begin == own_end == 0 means "the whole graph" */

/* scan orders (nsk_set_scan) */
#define NSK_SCAN_CHROMATIC 0    /* colour classes in parallel, Philox uniforms (default)          */
#define NSK_SCAN_SEQUENTIAL 1   /* one lane, variable-id order, MT19937: the reference's own
                                   trajectory (validation only; slow)                             */

/* The arrays a FactorGraph is constructed from (factorgraph.py:30-37). */
typedef struct {
    int64_t nweight, nvar, nfactor, nedge, nvtf, nfactor_index;
    const nsk_weight *weight;
    const nsk_variable *variable;
    const nsk_factor *factor;
    const nsk_ftv *fmap;
    const nsk_vtf *vmap;
    const int64_t *factor_index;
    int32_t flags;
    int32_t device;        /* HIP device ordinal                                                  */
    int64_t own_begin;     /* this handle samples variables [own_begin, own_end); the rest are     */
    int64_t own_end;       /* ghosts (= the reference's isEvidence==4, inference.py:21-23)         */
} nsk_graph_desc;

typedef struct nsk_graph nsk_graph;

/* Replaces FactorGraph.__init__'s state setup (factorgraph.py:39-63): validates the graph
 * (unknown factor functions -> NSK_E_FACTOR_FUNC, out-of-range indices -> NSK_E_INDEX), colours
 * it, compiles the device layout and uploads it.  State starts as the reference's: values and
 * evidence-chain values = initialValue, weights = initialValue, counts = 0. */
int nsk_graph_create(const nsk_graph_desc *desc, nsk_graph **out);
int nsk_graph_destroy(nsk_graph *g);

/* Host <-> device state sync.  Arrays are the FactorGraph attributes var_value[0],
 * var_value_evid[0], weight_value[0], count (factorgraph.py:46-53); NULL = leave alone. */
int nsk_state_upload(nsk_graph *g, const int64_t *var_value, const int64_t *var_value_evid,
                     const double *weight_value, const int64_t *count);
int nsk_state_download(nsk_graph *g, int64_t *var_value, int64_t *var_value_evid,
                       double *weight_value, int64_t *count);

/* RNG: the chromatic scan draws from Philox4x32-10 keyed by `seed`.  A variable's generator id is
 * its position in the compiled layout (nsk_graph_get_layout), so samples are a function of the seed
 * AND the layout the library chose (device, flags and diagnostic switches being equal, a graph
 * always compiles to the same layout).  Learning sweeps: counter (id, stream, sweep index);
 * inference sweeps: ids q and q + 64 with equal q >> 7 share one block, counter
 * ((q >> 7) * 64 + (q & 63), 0, sweep index), words 0-1 / 2-3 (pair scheme), except inside segments with
 * draw tables, where four ids share two blocks (quad and wide schemes, nsk_graph_get_generators).  The sweep index starts at `sweep0`
 * and advances by one per sweep of any kind; its high half (counter word 3) is XORed with the
 * handle's shard tag (nsk_set_rng_tag; 0 for a handle that owns the whole graph).  Sequential scan seeds MT19937 like
 * np.random.seed(seed); random.seed(seed). */
int nsk_set_seed(nsk_graph *g, uint64_t seed, uint64_t sweep0);
/* Shards of one graph must not share generator streams although their generator ids (layout
 * positions) coincide: Philox counter word 3 is the sweep index's high half XOR a shard tag.  The
 * tag defaults to own_begin (distinct for disjoint shards of one variable array); a handle built
 * from a shard-local graph (own variables + ghosts, renumbered) passes the GLOBAL id of its first
 * owned variable here, which also makes it sample exactly what the whole-graph handle with that
 * own_range samples. */
int nsk_set_rng_tag(nsk_graph *g, uint32_t tag);
int nsk_set_scan(nsk_graph *g, int scan);
/* Chromatic learning applies sample_and_sgd's update (learning.py:110-125) once per colour class:
 * a weight visited k times moves by step * (sum of its k gradients).  The reference updates per
 * visit, which is stable for any k * step; the batch is not, so a weight with k * step > cap uses
 * step = cap / k in that class (same fixed point).  Default 0.5; cap <= 0 switches clipping off.
 * nsk_graph_info.learn_clipped counts the clipped updates. */
int nsk_set_learn_cap(nsk_graph *g, double cap);
/* Chromatic learning, one-class lag (default on): the weight update of colour class c is applied beside
 * the sampling of class c + 1 (it rides in block 0 of that class's launch), which therefore sees the weights
 * as of the end of class c - 1 -- milder than the staleness of the reference's own Hogwild threads and of its distributed merge
 * (one epoch, salt/src/numbskull_master.py:223-224).  The pipeline is drained at the end of every
 * nsk_learn_sweeps call (the weights it leaves include every update).  It applies to handles with at most
 * 256 weights (they accumulate in LDS and the update rides in the next class's launch: the grids); with a
 * larger weight table the update is a launch of its own either way and every class waits for it
 * (nsk_graph_info.learn_lag says which).  lag = 0: every class waits for the previous class's update.  Same
 * fixed point either way; the oracle's device mode mirrors both. */
int nsk_set_learn_lag(nsk_graph *g, int lag);

/* Replaces run_pool(gibbsthread) at factorgraph.py:141 (burnin=1) and :163 (burnin=0):
 * `nsweeps` epochs of gibbsthread (inference.py:10-33) over the owned variables. */
int nsk_gibbs_sweeps(nsk_graph *g, int64_t nsweeps, int sample_evidence, int burnin);

/* Replaces the epoch loop around run_pool(learnthread) at factorgraph.py:194-206: `nsweeps`
 * epochs of learnthread/sample_and_sgd (learning.py:12-125), step *= decay after each. */
int nsk_learn_sweeps(nsk_graph *g, int64_t nsweeps, double step, double decay,
                     int regularization, double reg_param, int64_t truncation,
                     int learn_non_evidence);

/* Introspection used by tests, bench and the multi-GPU host code. */
typedef struct {
    int64_t nvar, nowned, ncolors, value_bytes, device_bytes;
    int64_t nfast, ngeneric;          /* variables on the inlined / the generic kernel path       */
    double alg_bytes_inference;       /* algorithmic bytes per sweep       */
    double alg_bytes_learning;
    int64_t sweeps_done;
    double layout_bytes_inference;    /* bytes one sweep must move in the compiled device layout  */
    double layout_bytes_learning;     /*   (tile words, positions, distinct values, stores, tally) */
    int64_t ztab_entries;             /* draw-table entries (0: no tabulated program)              */
    double compile_seconds;           /* host time spent in the graph compiler                     */
    double learn_cap;                 /* nsk_set_learn_cap                                         */
    int64_t learn_clipped;            /* weight updates whose step was clipped so far              */
    int64_t grad_shift;               /* gradient sums are Q(31+s).(32-s) fixed point: s (0 unless one
                                         weight's sum in one colour class could reach 2^30)        */
    int64_t acc_copies;               /* copies of the global learning accumulators: 8 (one per XCD, after
                                         the device passed the self-test of nsk_graph_create), 1 for a
                                         graph that accumulates in LDS or a device that failed it; +16
                                         when the LDS path's bins are private to XCDs; 0 from
                                         nsk_graph_plan (no device)                                */
    int64_t learn_lag;                /* 1: learning sweeps of this handle run with the one-class lag
                                         (nsk_set_learn_lag on AND at most 256 weights: the update then
                                         rides in the next class's launch); 0: every class waits for the
                                         previous class's update                                    */
    int64_t direct_weights;           /* weights with a single factor that the learning kernels update in
                                         place at their one visit per class (no accumulator, no update
                                         launch for them); 0: none                                 */
    int64_t weight_slots;             /* 1: the device table keeps the single-factor weights in layout order
                                         (nsk_graph_get_weight_slots is not the identity); 0: caller's order */
    int64_t layout_hash;              /* with NSK_LAYOUT_HASH=1 in the environment: a 64-bit hash of every
                                         array of the compiled layout (colours, positions, tile and group
                                         streams, programs, weight slots ...) -- what a check that two builds /
                                         thread counts / library versions compile a graph alike compares;
                                         0 otherwise (hashing the layout of a 50M-variable graph takes seconds) */
    int64_t p2p_fused;                /* 1: after nsk_p2p_import(_local), the handle's inference sweeps exchange the
                                         boundary INSIDE their table launches (nsk_gibbs_sweeps_p2p: border tiles read
                                         the receive block, write into the readers' and raise the flags; no exchange
                                         kernels per sweep) -- a shard whose sampled variables all live in table
                                         segments and whose boundary values have one reader each; 0: exchange kernels */
    int64_t tab_quads;                /* position quads (256 consecutive positions) of the segments with draw tables */
    int64_t wide_quads;               /* ... of them the WIDE ones: one lane samples four consecutive positions from
                                         dword loads (nsk_graph_get_generators bit 41)                              */
} nsk_graph_info;
int nsk_graph_get_info(nsk_graph *g, nsk_graph_info *info);
int nsk_graph_get_colors(nsk_graph *g, int32_t *color /* nvar, -1 for ghosts */);
/* Internal numbering: the device keeps values in the order of its compiled layout (a sampled
 * variable at its position, the others after them).  iid[v] (nvar entries, may be NULL) = index of
 * variable v in the NSK_BUF_VALUE / NSK_BUF_VALUE_EVID buffers, *nid = their length in elements.
 * nsk_state_upload / nsk_state_download and the exchange lists take the caller's variable ids. */
int nsk_graph_get_layout(nsk_graph *g, int32_t *iid, int64_t *nid);
/* The chromatic scan's generator of every variable (nvar entries; -1: not sampled by this handle): bits
 * 0-39 the generator id (= the position in the compiled layout), bit 40 set when the variable's INFERENCE
 * draws come from the quad scheme -- positions inside segments with draw tables: ids q, q + 64, q + 128,
 * q + 192 with equal q >> 8 share two Philox blocks, counter ((q >> 8) * 64 + (q & 63), stream, sweep),
 * stream 2 word (q >> 6) & 3 = the draw's high word, stream 3 the same word its low word -- instead of
 * the pair scheme described at nsk_set_seed; bit 41 set instead when they come from the WIDE scheme -- positions
 * inside wide quads (nsk_graph_info.wide_quads: 256 consecutive positions of a table segment whose members are
 * consecutive positions too, so that one lane samples four of them from dword loads): ids 4 i .. 4 i + 3 with equal
 * q >> 2 share the two blocks, counter ((q >> 8) * 64 + ((q >> 2) & 63), stream, sweep), word q & 3 of stream 2 / 3 =
 * the draw's high / low word.  What a checker needs to reproduce the samples. */
int nsk_graph_get_generators(nsk_graph *g, int64_t *gen);

/* Weight slots: where the device table (NSK_BUF_WEIGHT) keeps each weight, slot[w] for the caller's id w
 * (nweight entries).  The identity except on whole-graph handles that update single-factor weights in place
 * (nsk_graph_info.direct_weights): there those weights are numbered among themselves in the order the layout
 * first meets them, so that the entries a workgroup walks read and update neighbouring slots.
 * nsk_state_upload / nsk_state_download always speak the caller's ids.  Returns 1 when some slot differs
 * from its id, 0 for the identity, < 0 on error. */
int nsk_graph_get_weight_slots(nsk_graph *g, int64_t *slot);

/* Host-only planning (no GPU touched): validate + colour the graph exactly as nsk_graph_create
 * would and report the colours / sizes.  `color` (nvar entries, may be NULL) gets -1 for variables
 * this handle does not sample. */
int nsk_graph_plan(const nsk_graph_desc *desc, int32_t *color, nsk_graph_info *info);
/* Host-only twin of nsk_ghost_needs (vids == NULL: query the count). */
int nsk_graph_plan_needs(const nsk_graph_desc *desc, int64_t *count, int32_t *vids);

/* HIP-event bracket on the library's stream: elapsed ms and number of sweep-kernel launches
 * between begin and end (bench.py's roofline leg). */
int nsk_profile_begin(nsk_graph *g);
int nsk_profile_end(nsk_graph *g, double *elapsed_ms, int64_t *kernel_launches);
/* the same bracket without blocking at its end: nsk_profile_mark records the closing event,
 * nsk_profile_read waits for it (handles whose streams wait for each other are marked first, read later) */
int nsk_profile_mark(nsk_graph *g);
int nsk_profile_read(nsk_graph *g, double *elapsed_ms, int64_t *kernel_launches);

/* Multi-GPU plumbing: raw device addresses of the value arrays (element size = value_bytes,
 * indexed by INTERNAL id, see nsk_graph_get_layout), of the weights (float64, indexed by weight SLOT: the
 * caller's id except where nsk_graph_get_weight_slots says otherwise -- never on a handle that samples a
 * range of a larger graph) and of the boundary staging buffers, and the stream the library launches on. */
#define NSK_BUF_VALUE 0
#define NSK_BUF_VALUE_EVID 1
#define NSK_BUF_WEIGHT 2
#define NSK_BUF_SEND 3         /* boundary exchange staging, after nsk_exchange_setup */
#define NSK_BUF_RECV 4
#define NSK_BUF_SEND_EVID 5
#define NSK_BUF_RECV_EVID 6
int nsk_device_buffer(nsk_graph *g, int which, void **ptr, int64_t *nbytes);
int nsk_set_stream(nsk_graph *g, void *hip_stream);

/* Boundary ("ghost") exchange for a range-partitioned graph: the reference copies owners' values to
 * the replicas once per epoch (salt/src/numbskull_master.py:165-224); here only the values another
 * partition actually reads travel.
 *   nsk_ghost_needs     sorted ids of the variables outside [own_begin, own_end) that this handle's
 *                       variables read (vids == NULL: query the count)
 *   nsk_exchange_setup  send_vids: the owned variables some other rank reads, in the order every
 *                       rank agreed on; recv_vids / recv_off: the same lists of all `world` ranks,
 *                       concatenated (-1: a variable this handle does not hold -- a shard-local graph
 *                       keeps only the ghosts it reads -- is skipped); slot: elements reserved per
 *                       rank in the gathered buffer
 *   nsk_exchange_pack   SEND[i] = value[send_vids[i]]          (which = NSK_BUF_VALUE[_EVID])
 *   nsk_exchange_unpack value[recv_vids[j]] = RECV[src*slot + j - recv_off[src]] for every src != rank
 * The all-gather of SEND into RECV is done by the caller (torch.distributed) or by the native loop
 * below. */
int nsk_ghost_needs(nsk_graph *g, int64_t *count, int32_t *vids);
int nsk_exchange_setup(nsk_graph *g, int world, int rank, const int32_t *send_vids, int64_t nsend,
                       const int32_t *recv_vids, const int64_t *recv_off, int64_t slot);
int nsk_exchange_pack(nsk_graph *g, int which);
int nsk_exchange_unpack(nsk_graph *g, int which);

/* Native RCCL loop: `nsweeps` x (sweep, pack, ncclAllGather over xGMI, unpack) enqueued on the
 * library's stream without returning to the host language.  unique_id: the 128-byte ncclUniqueId
 * created by nsk_comm_unique_id on rank 0 and broadcast by the caller; librccl_path: the RCCL
 * shared object to bind (the one torch has loaded). */
int nsk_comm_unique_id(const char *librccl_path, void *id128);
int nsk_comm_init(nsk_graph *g, int world, int rank, const void *id128, const char *librccl_path);
int nsk_gibbs_sweeps_exchange(nsk_graph *g, int64_t nsweeps, int sample_evidence, int burnin);
int nsk_learn_sweeps_exchange(nsk_graph *g, int64_t nsweeps, double step, double decay,
                              int regularization, double reg_param, int64_t truncation,
                              int learn_non_evidence);
int nsk_synchronize(nsk_graph *g);

/* Peer-to-peer boundary exchange on one node: instead of pack -> ncclAllGather -> unpack, a rank writes
 * the boundary values each peer reads straight into that peer's buffer (device memory the peer exposes
 * with hipIpc; over xGMI between GPUs) and raises a flag there; a rank waits for the flags of its peers
 * and scatters their values -- no collective, no host round trip per sweep (the reference's per-epoch
 * owner -> replica copy, salt/src/numbskull_master.py:165-224).  Learning epochs send both chains the same
 * way and merge the epoch's weight deltas as w = w_start + (d_0 + d_1 + ...) in rank order (the master's
 * rule, numbskull_master.py:223-224; minions' deltas numbskull_minion.py:260-280) by reduce-scatter +
 * all-gather over the same buffers: rank q owns slice [q nw / W, (q + 1) nw / W) of the weight vector, every
 * rank writes slice q of its deltas to rank q only, the owner adds them in rank order and writes the merged
 * slice to every rank -- one owner per weight, so all ranks end up with bit-identical weights.
 *   nsk_p2p_setup   PAIRWISE lists: send_vids[send_off[q] .. send_off[q+1]) = the owned variables rank q
 *                   reads, recv_vids[recv_off[q] ..) = the variables this handle reads from rank q, both in
 *                   the order the two sides agreed on (ascending global id); peer_base[q] = where this
 *                   rank's segment starts in q's receive list, peer_total[q] = length of q's receive list
 *   nsk_p2p_export  allocates this rank's buffer; handle64 (may be NULL) receives its hipIpc handle, *base
 *                   (may be NULL) its device address
 *   nsk_p2p_import  all_handles: world x 64 bytes, gathered by the caller (one process per rank)
 *   nsk_p2p_import_local  bases[q] = device address of rank q's buffer, for ranks that live in ONE process
 *                   (several handles on one device, or one process driving several GPUs with peer access)
 *   nsk_gibbs_sweeps_p2p / nsk_learn_sweeps_p2p  `nsweeps` x (sweep, push to peers, wait + unpack [+ weight
 *                   merge]) enqueued on the library's stream; asynchronous like nsk_gibbs_sweeps
 *   nsk_p2p_exchange  one exchange alone (learn != 0: both chains + weight deltas); part 0 = all of it,
 *                   1 = the pushes, 2 = wait + unpack (+ the owner's half of the weight merge), 3 = the closing
 *                   half of the weight merge (phase timings; several handles driven by one process issue the
 *                   parts breadth-first)
 *   nsk_p2p_selftest  the same kernels with a payload pattern that depends on sender, element, exchange and
 *                   chain instead of the state, compared on the receiving side instead of stored (state and
 *                   weights are left alone): checks that peer WRITES are visible, not only the flags; a
 *                   mismatch is reported by nsk_p2p_check.  Same `learn` / `part` arguments
 *   nsk_p2p_check   synchronises and returns NSK_E_DEVICE when a peer's flag did not arrive within
 *                   NSK_P2P_TIMEOUT_S seconds (default 30), or a self-test payload differed, since the last
 *                   check (nsk_state_download reports a time-out too) */
/* Partial factors (salt/src/messages.py:1333-1355 compute_pf_values / apply_pf_values): a reader shard that holds a
 * factor OR / AND / ISTRUE with several members owned by THIS shard takes ONE aggregate of them instead of every
 * member -- op 0: "some member is 1" (OR), op 1: "no member is 0" (AND, ISTRUE), a 1 / 0 value the reader keeps in a
 * boolean ghost variable that stands in the factor for those members (numbskull_amd/graphgen.py partial_factors does
 * the rewriting; the factor's value, hence every sample, is exactly what it is with the members themselves).
 * nsk_pf_setup registers the aggregates this handle computes: members of aggregate j = member_vids[member_off[j] ..
 * member_off[j+1]) (variables the handle holds).  A send list of nsk_p2p_setup (call it afterwards) names aggregate j
 * as variable id nvar + j; every peer-to-peer exchange recomputes them (both chains in learning) before it pushes. */
int nsk_pf_setup(nsk_graph *g, int64_t npf, const uint8_t *op, const int64_t *member_off, const int32_t *member_vids);
int nsk_p2p_setup(nsk_graph *g, int world, int rank, const int32_t *send_vids, const int64_t *send_off,
                  const int32_t *recv_vids, const int64_t *recv_off, const int64_t *peer_base,
                  const int64_t *peer_total);
int nsk_p2p_export(nsk_graph *g, void *handle64, void **base);
int nsk_p2p_import(nsk_graph *g, const void *all_handles);
int nsk_p2p_import_local(nsk_graph *g, void *const *bases);
int nsk_gibbs_sweeps_p2p(nsk_graph *g, int64_t nsweeps, int sample_evidence, int burnin);
int nsk_learn_sweeps_p2p(nsk_graph *g, int64_t nsweeps, double step, double decay, int regularization,
                         double reg_param, int64_t truncation, int learn_non_evidence);
int nsk_p2p_exchange(nsk_graph *g, int learn, int part);
int nsk_p2p_selftest(nsk_graph *g, int learn, int part);
/* The fused exchange (nsk_graph_info.p2p_fused) reads and writes peer memory with system-coherent loads / stores
 * and no fences.  nsk_p2p_selftest(g, 2, part) runs that protocol in isolation with a pattern payload (part 0: all
 * of it; 1 / 2: the writing / the reading half); nsk_p2p_fuse(g, 0) makes the handle keep the exchange kernels
 * (what PartitionedSampler does on every rank when any rank's test fails), nsk_p2p_fuse(g, 1) plans the fused
 * exchange again.  Returns 1 / 0 = fused or not afterwards, < 0 on error. */
int nsk_p2p_fuse(nsk_graph *g, int on);
int nsk_p2p_check(nsk_graph *g);
/* After a failed self-test the peers have advanced their exchange tags and written patterns into this rank's receive
 * blocks: nsk_p2p_reset zeroes this rank's exchange allocation (flags, both parities of the receive blocks, the
 * weight-delta slices), its error words and its tag.  EVERY rank calls it, with a barrier of the caller's before and
 * after (no peer may still be writing; none may start before all are clean): the mappings stay. */
int nsk_p2p_reset(nsk_graph *g);

/* ---- graph-aware partitioning (no GPU needed; salt/src/messages.py:542-670 find_connected_components /
 * find_metis_parts).  A partition is a VARIABLE ORDER in front of the samplers' range partition
 * (inference.py:17-18): nsk_graph_order keeps connected components together (method 0) and walks every component
 * breadth first from a pseudo-peripheral variable (method 1, Cuthill-McKee), so that the shard formula's cut runs
 * along a few BFS fronts; order[new id] = old id, cc_id[old id] = connected component (may be NULL), *ncc their
 * number.  nsk_comm_volume: the communication volume -- (variable, foreign part) pairs read across the cut, i.e.
 * the values one exchange moves, METIS' objtype = vol -- of the range partition into nparts <= 64 shards of the ids
 * new_id[old id] (NULL: the caller's ids). ---- */
int nsk_graph_order(int64_t nvar, int64_t nfactor, const nsk_factor *factor, int64_t nedge, const nsk_ftv *fmap,
                    int method, int64_t *order, int64_t *cc_id, int64_t *ncc);
int nsk_comm_volume(int64_t nvar, int64_t nfactor, const nsk_factor *factor, int64_t nedge, const nsk_ftv *fmap,
                    const int64_t *new_id, int nparts, int64_t *volume);
/* The multilevel partitioner (find_metis_parts, messages.py:593-670: nxmetis.partition with objtype = vol): heavy-edge
 * matching, a partition of the coarsest graph, greedy k-way boundary refinement on the way up and, at the finest
 * level, refinement of nsk_comm_volume itself with the parts held at exactly the shard formula's sizes.
 * order[new id] = old id: range g of the new ids is part g; part[old id] = g (may be NULL); stats (may be NULL)
 * receives 5 numbers: levels, coarsest vertices, edge cut after uncoarsening, communication volume before / after
 * the last refinement.  Deterministic in (graph, nparts, seed); nparts <= 64, nvar < 2^31. */
int nsk_graph_partition(int64_t nvar, int64_t nfactor, const nsk_factor *factor, int64_t nedge, const nsk_ftv *fmap,
                        int nparts, uint64_t seed, int64_t *order, int64_t *part, int64_t *stats);

/* ---- host-side index build and file parsing (no GPU needed) ---- */
/* dataloading.compute_var_map (dataloading.py:16-81), native and O(edges). */
int nsk_compute_var_map(int64_t nvar, const nsk_variable *variable, int64_t nfactor,
                        const nsk_factor *factor, int64_t nedge, const nsk_ftv *fmap,
                        int64_t nvtf, nsk_vtf *vmap, int64_t nfactor_index, int64_t *factor_index,
                        const uint8_t *domain_mask, const int64_t *factors_to_skip, int64_t nskip);
/* FactorGraph.__init__'s arrays (factorgraph.py:41-53) from the packed records, over the host threads: cstart[nvar + 1]
 * (cumulative tally slots: one for a binary variable, `cardinality` otherwise), init[nvar] (initialValue, dense),
 * *max_card, *longest (the longest factor_index_length of vmap; both may be NULL). */
int nsk_state_layout(int64_t nvar, const nsk_variable *variable, int64_t nvtf, const nsk_vtf *vmap,
                     int64_t *cstart, int64_t *init, int64_t *max_card, int64_t *longest);
/* dataloading.load_factors (dataloading.py:196-235) on the raw bytes of graph.factors. */
int nsk_parse_factors(const uint8_t *data, int64_t nbytes, int64_t nfactor, int64_t nedge,
                      nsk_factor *factor, nsk_ftv *fmap, const uint8_t *domain_mask,
                      const nsk_variable *variable, int64_t nvar, const nsk_vtf *vmap);

/* dataloading.load_domains (dataloading.py:159-187) on the raw bytes of graph.domains: marks
 * domain_mask, fills vmap[vtf_offset ..].value and re-maps initialValue exactly as the reference. */
int nsk_parse_domains(const uint8_t *data, int64_t nbytes, uint8_t *domain_mask, int64_t nvar,
                      nsk_variable *variable, nsk_vtf *vmap, int64_t nvtf);
/* FactorGraph.dump_probabilities (factorgraph.py:216-229): the "<vid> <value> <prob>" text file. */
int nsk_write_probabilities(const char *path, int64_t nvar, const nsk_variable *variable,
                            const nsk_vtf *vmap, int64_t nvtf, const int64_t *cstart,
                            const int64_t *count, int64_t ncount, double epochs);

/* Self-test hooks: run the device exp / Philox on caller data (parity tests vs the oracle). */
int nsk_selftest_exp(int device, const double *x, double *y, int64_t n);
int nsk_selftest_philox(int device, uint64_t seed, uint64_t sweep, uint32_t stream, int64_t n,
                        uint32_t *out /* 4*n words, counter c0 = 0..n-1 */);

/* Plain HBM stream-copy of `nbytes` (read + write) with `width`-byte accesses per lane (4 or 16;
 * 64 = four 16-byte non-temporal loads in flight per lane and non-temporal stores, the ceiling):
 * the achievable-bandwidth ceiling bench.py reports beside the sweep, and the calibration
 * workload for the rocprofv3 FETCH_SIZE / WRITE_SIZE counters. */
int nsk_selftest_stream(int device, int64_t nbytes, int width, int iters, double *gbytes_per_s);

int nsk_device_count(int *count);
const char *nsk_last_error(void);
const char *nsk_version(void);

#ifdef __cplusplus
}
#endif
#endif
