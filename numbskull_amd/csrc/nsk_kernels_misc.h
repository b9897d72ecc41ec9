// nsk_kernels_misc.h -- tally folding, boundary exchange, sequential validation scan and self-test
// kernels.
#pragma once

#include "nsk_device.h"
#include "nsk_kernels_gibbs.h"

namespace nsk {

// int32 per-call tally deltas -> int64 master copy (the host-visible `count`)
static __global__ __launch_bounds__(NSK_BLOCK) void k_fold_counts(int32_t *delta, long long *total, int n) {
    const int i = (int)(blockIdx.x * NSK_BLOCK + threadIdx.x);
    if (i >= n) return;
    total[i] += (long long)delta[i];
    delta[i] = 0;
}

template <typename T>
__global__ __launch_bounds__(NSK_BLOCK) void k_stream_copy(const T *__restrict__ src, T *__restrict__ dst,
                                                           long long n) {
    const long long stride = (long long)gridDim.x * NSK_BLOCK;
    for (long long i = (long long)blockIdx.x * NSK_BLOCK + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

// achievable-bandwidth ceiling: every lane moves UNR x 16 bytes per trip, all loads issued before
// the stores, non-temporal both ways (the data is touched once)
template <int UNR>
__global__ __launch_bounds__(NSK_BLOCK) void k_stream_copy_nt(const uint4 *__restrict__ src, uint4 *__restrict__ dst,
                                                              long long n) {
    const long long stride = (long long)gridDim.x * NSK_BLOCK;
    long long i = (long long)blockIdx.x * NSK_BLOCK + threadIdx.x;
    for (; i + (UNR - 1) * stride < n; i += UNR * stride) {
        nsk_u32x4 v[UNR];
#pragma unroll
        for (int k = 0; k < UNR; k++) v[k] = __builtin_nontemporal_load((const nsk_u32x4 *)(src + i + k * stride));
#pragma unroll
        for (int k = 0; k < UNR; k++) __builtin_nontemporal_store(v[k], (nsk_u32x4 *)(dst + i + k * stride));
    }
    for (; i < n; i += stride) dst[i] = src[i];
}

// ---- state transfer: caller's order (variable id) <-> internal order (layout position) -----------------
template <typename VT>
__global__ __launch_bounds__(NSK_BLOCK) void k_state_scatter(VT *val, const int32_t *iid, const VT *by_vid, long long nvar) {
    for (long long v = (long long)blockIdx.x * NSK_BLOCK + threadIdx.x; v < nvar; v += (long long)gridDim.x * NSK_BLOCK)
        val[iid[v]] = by_vid[v];
}
template <typename VT>
__global__ __launch_bounds__(NSK_BLOCK) void k_state_gather(const VT *val, const int32_t *iid, VT *by_vid, long long nvar) {
    for (long long v = (long long)blockIdx.x * NSK_BLOCK + threadIdx.x; v < nvar; v += (long long)gridDim.x * NSK_BLOCK)
        by_vid[v] = val[iid[v]];
}

// the int64 tally narrowed to int32 for the trip over PCIe (flag: some count does not fit)
static __global__ __launch_bounds__(NSK_BLOCK) void k_count_narrow(const long long *total, int32_t *out, long long n, unsigned int *wide) {
    for (long long i = (long long)blockIdx.x * NSK_BLOCK + threadIdx.x; i < n; i += (long long)gridDim.x * NSK_BLOCK) {
        const long long x = total[i];
        if (x < INT32_MIN || x > INT32_MAX) *wide = 1u;
        out[i] = (int32_t)x;
    }
}

// ---- boundary exchange ---------------------------------------------------------------------------
template <typename VT>
__global__ __launch_bounds__(NSK_BLOCK) void k_exchange_pack(const VT *val, const int32_t *send_vids,
                                                             VT *sendbuf, int n) {
    const int i = (int)(blockIdx.x * NSK_BLOCK + threadIdx.x);
    if (i < n) sendbuf[i] = val[send_vids[i]];
}

// recv_src[j] = rank that owns recv_vids[j]; entries of this rank itself are skipped
template <typename VT>
__global__ __launch_bounds__(NSK_BLOCK) void k_exchange_unpack(VT *val, const int32_t *recv_vids,
                                                               const int32_t *recv_slot, const VT *recvbuf,
                                                               int n) {
    const int j = (int)(blockIdx.x * NSK_BLOCK + threadIdx.x);
    if (j >= n) return;
    const int sl = recv_slot[j];            // index into the gathered buffer, -1 = own entry
    if (sl >= 0) val[recv_vids[j]] = recvbuf[sl];
}

// ---- peer-to-peer boundary exchange (one node: xGMI between GPUs / one device shared by ranks) ------
// Every rank owns ONE fine-grained allocation its peers have mapped (hipIpc, or plain pointers when the
// ranks live in one process):
//     flags[2][2][world]                  per parity: [0] tag of the last exchange whose VALUES (and weight-delta
//                                         slices) rank q has written here, [1] ... whose MERGED weight slice
//     recv[2][2][nrecv]                   per parity, per chain (var_value, var_value_evid): the values this
//                                         rank reads from others, segment of source q at roff[q]
//     sbuf[2][world][slice_max]           per parity: every rank's weight deltas of a learning epoch FOR THE
//                                         SLICE OF THE WEIGHT VECTOR THIS RANK OWNS (rank q owns [q nw / W, (q+1) nw / W))
//     gbuf[2][nweight]                    per parity: the merged weights, slice q written by its owner q
// Boundary lists are PAIRWISE: rank s sends rank d exactly the variables d reads from s (ascending global
// id on both sides), so nothing travels that its receiver does not read.  After a sweep a rank WRITES its
// boundary values straight into the peers' buffers; the last block to finish raises this rank's flag at every
// peer; a rank waits for the flags of its peers and scatters their values.  No collective, no host in the
// loop.  Peers are symmetric (q is a peer when either side reads from the other), so two ranks that exchange
// anything wait for each other in every exchange and a peer runs at most one exchange ahead: two parities
// suffice.
// A learning epoch's weight deltas travel as REDUCE-SCATTER + ALL-GATHER over the same buffers (SURVEY 8(e);
// the master's rule w = w_start + sum of deltas, salt/src/numbskull_master.py:223-224, numbskull_minion.py:
// 270-279): with the values, every rank writes slice q of its deltas to rank q only; the owner adds the W
// contributions IN RANK ORDER, forms w_start + sum and writes the merged slice to every rank; a rank waits
// for the W owners' second flags and takes the merged vector -- 2 x 7 x nw / 8 doubles per rank and epoch
// instead of 7 x nw, the same additions in the same order on ONE owner, hence bit-identical weights on all.
struct P2PPlan {                        // (one node: at most 16 ranks)
    void *base[16];                     // peer q's allocation
    unsigned long long soff[17];        // this rank's send list = concatenation over q of what q reads: [soff[q], soff[q+1])
    unsigned long long dbase[16];       // where this rank's segment starts inside q's per-chain block
    unsigned long long dtotal[16];      // elements of q's per-chain block (q's receive total)
    unsigned long long roff[17];        // this rank's receive block: segment of source q = [roff[q], roff[q+1])
};
#define NSK_P2P_ALIGN 256ull
#define NSK_P2P_ERR_TIMEOUT 1u
#define NSK_P2P_ERR_PAYLOAD 2u
__host__ __device__ inline size_t nsk_p2p_align(size_t x) { return (x + NSK_P2P_ALIGN - 1) / NSK_P2P_ALIGN * NSK_P2P_ALIGN; }
__host__ __device__ inline size_t nsk_p2p_recv_off(int world) {           // byte offset of recv[] in an allocation
    return nsk_p2p_align((size_t)(4 * world) * sizeof(unsigned int));
}
__host__ __device__ inline size_t nsk_p2p_sbuf_off(int world, size_t nrecv, size_t vbytes) {
    return nsk_p2p_recv_off(world) + nsk_p2p_align(4 * nrecv * vbytes);
}
__host__ __device__ inline size_t nsk_p2p_slice_lo(int q, int world, size_t nw) { return (size_t)q * nw / (size_t)world; }
__host__ __device__ inline size_t nsk_p2p_slice_max(int world, size_t nw) { return (nw + (size_t)world - 1) / (size_t)world; }
__host__ __device__ inline size_t nsk_p2p_gbuf_off(int world, size_t nrecv, size_t vbytes, size_t nw) {
    return nsk_p2p_sbuf_off(world, nrecv, vbytes) + nsk_p2p_align(2 * (size_t)world * nsk_p2p_slice_max(world, nw) * sizeof(double));
}
__host__ __device__ inline size_t nsk_p2p_bytes(int world, size_t nrecv, size_t vbytes, size_t nw) {
    return nsk_p2p_gbuf_off(world, nrecv, vbytes, nw) + nsk_p2p_align(2 * nw * sizeof(double)) + 256;
}

// The set-up self-test runs the very kernels of the sweep loops with `selftest` set: they then push a pattern
// that depends on the sender, the element, the exchange tag and the chain instead of the state, and the
// receiving side compares instead of storing -- payload visibility across devices is checked, not only the flags.
template <typename VT>
__device__ __forceinline__ VT p2p_pattern(int src, unsigned long long k, unsigned int tag, int chain) {
    return (VT)(((unsigned int)(src * 37 + 1) * 7u + (unsigned int)k * 11u + tag * 5u + (unsigned int)chain * 3u) & 127u);
}
__device__ __forceinline__ double p2p_pattern_dw(int src, int i, unsigned int tag) {
    return (double)((src + 1) * 1000 + (i % 997)) * 0.25 + (double)(tag & 255u);
}

// weights of a learning exchange (null w: an inference exchange)
struct P2PWeights {
    const double *w, *w_start;          // this rank's weights now / at the start of the epoch
    int nw;
};

// push: boundary values into the readers' receive blocks, and (learning) slice q of the weight deltas into
// block `me` of rank q's sbuf -- this rank's own slice included
template <typename VT>
__device__ __forceinline__ void p2p_push(const VT *val, const VT *val_evid, int both, const int32_t *send_iid, long long nsend,
                                         const P2PPlan &pl, const P2PWeights &pw, int world, int me, unsigned int tag, int selftest) {
    const size_t par = tag & 1u;
    const size_t roff = nsk_p2p_recv_off(world);
    const long long stride = (long long)gridDim.x * NSK_BLOCK;
    for (long long k = (long long)blockIdx.x * NSK_BLOCK + threadIdx.x; k < nsend; k += stride) {
        int q = 0;
        while (q + 1 < world && (unsigned long long)k >= pl.soff[q + 1]) q++;
        const size_t tot = (size_t)pl.dtotal[q];
        const unsigned long long kl = (unsigned long long)k - pl.soff[q];
        VT *dst = (VT *)((char *)pl.base[q] + roff) + par * 2 * tot + (size_t)pl.dbase[q] + (size_t)kl;
        const int id = send_iid[k];
        dst[0] = selftest ? p2p_pattern<VT>(me, kl, tag, 0) : val[id];
        if (both) dst[tot] = selftest ? p2p_pattern<VT>(me, kl, tag, 1) : val_evid[id];
    }
    if (pw.w) {
        const size_t nw = (size_t)pw.nw, smax = nsk_p2p_slice_max(world, nw);
        for (long long i = (long long)blockIdx.x * NSK_BLOCK + threadIdx.x; i < (long long)nw; i += stride) {
            int q = (int)(((size_t)i * (size_t)world + (size_t)world - 1) / nw);          // owner of weight i: the q with lo(q) <= i < lo(q + 1)
            while (q > 0 && nsk_p2p_slice_lo(q, world, nw) > (size_t)i) q--;
            while (q + 1 < world && nsk_p2p_slice_lo(q + 1, world, nw) <= (size_t)i) q++;
            double *sb = (double *)((char *)pl.base[q] + nsk_p2p_sbuf_off(world, (size_t)pl.dtotal[q], sizeof(VT)));
            sb[(par * (size_t)world + (size_t)me) * smax + ((size_t)i - nsk_p2p_slice_lo(q, world, nw))] =
                selftest ? p2p_pattern_dw(me, (int)i, tag) : pw.w[i] - pw.w_start[i];
        }
    }
}

// true in every thread of the LAST block of the launch to get here (a handful of blocks: the ticket adds do not
// queue up); everything the launch -- and earlier kernels on the stream -- wrote is then visible to a peer that
// sees a flag raised afterwards
__device__ __forceinline__ bool p2p_last_block(unsigned int *ticket) {
    __shared__ unsigned int last;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    const bool l = last != 0u;
    if (l && threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return l;
}
__device__ __forceinline__ void p2p_raise(const P2PPlan &pl, int kind, int world, int me, unsigned int peer_mask, unsigned int tag) {
    __threadfence_system();
    if (threadIdx.x < (unsigned)world && ((peer_mask >> threadIdx.x) & 1u))
        __hip_atomic_store((unsigned int *)pl.base[threadIdx.x] + ((size_t)(tag & 1u) * 2 + (size_t)kind) * (size_t)world + (size_t)me, tag,
                           __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// waits (bounded: timeout_ticks of the 100 MHz wall clock, then *err |= time-out) for the tags of the peers;
// block-uniform result; an acquire fence at system scope follows (the payload was written by another agent)
__device__ __forceinline__ bool p2p_wait(const void *mine, int kind, int world, unsigned int peer_mask, unsigned int tag,
                                         unsigned int *err, unsigned long long timeout_ticks) {
    __shared__ int ok;
    __syncthreads();
    if (threadIdx.x == 0) {
        ok = 1;
        const unsigned long long t0 = wall_clock64();
        const unsigned int *flags = (const unsigned int *)mine + ((size_t)(tag & 1u) * 2 + (size_t)kind) * (size_t)world;
        for (int q = 0; q < world && ok; q++) {
            if (!((peer_mask >> q) & 1u)) continue;
            // (relaxed polls, ONE acquire fence behind the wait: an acquire load at system scope invalidates the
            // caches at every iteration -- with a waiting thread per block of a 391-block launch that was 90 us)
            while (__hip_atomic_load(flags + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != tag) {
                if (wall_clock64() - t0 > timeout_ticks) { ok = 0; break; }
                __builtin_amdgcn_s_sleep(8);
            }
        }
        if (!ok) (void)__hip_atomic_fetch_or(err, NSK_P2P_ERR_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const bool r = ok != 0;
    if (r) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    return r;
}
template <typename VT>
__device__ __forceinline__ void p2p_unpack(VT *val, VT *val_evid, int both, const int32_t *recv_iid, long long nrecv, const void *mine,
                                           const P2PPlan &pl, int world, unsigned int tag, int selftest, unsigned int *err) {
    const VT *rb = (const VT *)((const char *)mine + nsk_p2p_recv_off(world)) + (size_t)(tag & 1u) * 2 * (size_t)nrecv;
    for (long long j = (long long)blockIdx.x * NSK_BLOCK + threadIdx.x; j < nrecv; j += (long long)gridDim.x * NSK_BLOCK) {
        const VT a = __builtin_nontemporal_load(rb + j);
        const VT b = both ? __builtin_nontemporal_load(rb + (size_t)nrecv + j) : (VT)0;
        if (selftest) {
            int q = 0;
            while (q + 1 < world && (unsigned long long)j >= pl.roff[q + 1]) q++;
            const unsigned long long jl = (unsigned long long)j - pl.roff[q];
            if (a != p2p_pattern<VT>(q, jl, tag, 0) || (both && b != p2p_pattern<VT>(q, jl, tag, 1)))
                (void)__hip_atomic_fetch_or(err, NSK_P2P_ERR_PAYLOAD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;
        }
        const int id = recv_iid[j];
        val[id] = a;
        if (both) val_evid[id] = b;
    }
}
// the owner's half of the weight merge: this rank's slice = w_start + (d_0 + d_1 + ...) in rank order, written
// to the gbuf of every rank (its own included)
template <typename VT>
__device__ __forceinline__ void p2p_reduce_slice(const void *mine, long long nrecv, const P2PPlan &pl, const P2PWeights &pw,
                                                 int world, int me, unsigned int tag, int selftest) {
    const size_t nw = (size_t)pw.nw, smax = nsk_p2p_slice_max(world, nw), par = tag & 1u;
    const size_t lo = nsk_p2p_slice_lo(me, world, nw), hi = nsk_p2p_slice_lo(me + 1, world, nw);
    const double *sb = (const double *)((const char *)mine + nsk_p2p_sbuf_off(world, (size_t)nrecv, sizeof(VT))) + par * (size_t)world * smax;
    for (size_t i = lo + (size_t)blockIdx.x * NSK_BLOCK + threadIdx.x; i < hi; i += (size_t)gridDim.x * NSK_BLOCK) {
        // rank order, as ever; this rank's OWN contribution is computed here, not read from row `me` of sbuf: in the
        // single-launch exchange that row is written by other blocks of the same launch (p2p_push deals weight i to
        // another thread than this loop does), and a block waits for the PEERS' flags only -- the same double either
        // way, so the sum is bit-identical, without the read-after-write across blocks
        double t = 0.0;
        for (int r = 0; r < world; r++) {
            const double dr = r == me ? (selftest ? p2p_pattern_dw(me, (int)i, tag) : pw.w[i] - pw.w_start[i])
                                      : __builtin_nontemporal_load(sb + (size_t)r * smax + (i - lo));
            t = r == 0 ? dr : t + dr;
        }
        const double x = (selftest ? 0.0 : pw.w_start[i]) + t;
        for (int q = 0; q < world; q++) {
            double *gb = (double *)((char *)pl.base[q] + nsk_p2p_gbuf_off(world, (size_t)pl.dtotal[q], sizeof(VT), nw));
            gb[par * nw + i] = x;
        }
    }
}

// the pushes alone (tests that drive several handles from one process, phase timings)
template <typename VT>
__global__ __launch_bounds__(NSK_BLOCK) void k_p2p_push(const VT *val, const VT *val_evid, int both, const int32_t *send_iid,
                                                        long long nsend, P2PPlan pl, P2PWeights pw, int world, int me,
                                                        unsigned int peer_mask, unsigned int *ticket, unsigned int tag,
                                                        const unsigned long long *tag_base, int selftest) {
    if (tag_base) tag += (unsigned int)tag_base[1];        // captured launch: tag = device counter + offset
    p2p_push<VT>(val, val_evid, both, send_iid, nsend, pl, pw, world, me, tag, selftest);
    if (p2p_last_block(ticket)) p2p_raise(pl, 0, world, me, peer_mask, tag);
}

// One exchange in ONE launch (what the sweep loops enqueue): every block pushes its share, the last one to
// finish raises the flags, then every block goes on to wait for the peers' flags and unpacks its share --
// the wait depends on the peers' pushes only, so no block waits for another block of this launch.  A shard
// of the metric config is a few 4 us kernels per sweep: one kernel less per sweep is worth having.
// Learning: the blocks then add up this rank's slice of the weight deltas and write the merged slice to
// every rank; the last block to finish raises the second flag (k_p2p_gather_w waits for those).
// PUSH = false: the wait / unpack / reduce half alone.
template <typename VT, bool PUSH>
__global__ __launch_bounds__(NSK_BLOCK) void k_p2p_exchange(VT *val, VT *val_evid, int both, const int32_t *send_iid, long long nsend,
                                                            P2PPlan pl, P2PWeights pw, const int32_t *recv_iid, long long nrecv,
                                                            const void *mine, int world, int me, unsigned int peer_mask,
                                                            unsigned int *ticket, unsigned int tag, unsigned int *err,
                                                            const unsigned long long *tag_base, unsigned long long timeout_ticks,
                                                            int selftest) {
    if (tag_base) tag += (unsigned int)tag_base[1];        // captured launch: tag = device counter + offset
    if (PUSH) {
        p2p_push<VT>(val, val_evid, both, send_iid, nsend, pl, pw, world, me, tag, selftest);
        if (p2p_last_block(ticket)) p2p_raise(pl, 0, world, me, peer_mask, tag);
    }
    if (!p2p_wait(mine, 0, world, peer_mask, tag, err, timeout_ticks)) return;
    p2p_unpack<VT>(val, val_evid, both, recv_iid, nrecv, mine, pl, world, tag, selftest, err);
    if (pw.w) {
        p2p_reduce_slice<VT>(mine, nrecv, pl, pw, world, me, tag, selftest);
        if (p2p_last_block(ticket + 1)) p2p_raise(pl, 1, world, me, peer_mask, tag);
    }
}

// Large exchanges (send / receive lists beyond 2^16 values, weight tables beyond 2^16 weights: the 50M LR graph's
// shards push 490 000 values of two chains and 8 MB of weight deltas): launches with as many blocks as the lists
// want and NO ticket, fence or poll inside -- a one-wave k_p2p_raise / k_p2p_wait between them does those, and the
// stream's order does the rest (a kernel has its stores acknowledged before the next one starts).  Inside the
// exchange kernels' <= 64 blocks (their closing tickets, each behind a system-scope fence, must not queue up) the
// lists are latency-bound loops: 30 dependent trips per thread, 155 + 99 us per exchange on those shards.
template <typename VT>
__global__ __launch_bounds__(NSK_BLOCK) void k_p2p_push_big(const VT *val, const VT *val_evid, int both, const int32_t *send_iid,
                                                            long long nsend, P2PPlan pl, P2PWeights pw, int world, int me, unsigned int tag,
                                                            const unsigned long long *tag_base, int selftest) {
    if (tag_base) tag += (unsigned int)tag_base[1];
    p2p_push<VT>(val, val_evid, both, send_iid, nsend, pl, pw, world, me, tag, selftest);
}
template <typename VT>
__global__ __launch_bounds__(NSK_BLOCK) void k_p2p_unpack_big(VT *val, VT *val_evid, int both, const int32_t *recv_iid, long long nrecv,
                                                              const void *mine, P2PPlan pl, P2PWeights pw, int world, int me, unsigned int tag,
                                                              const unsigned long long *tag_base, unsigned int *err, int selftest) {
    if (tag_base) tag += (unsigned int)tag_base[1];
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & NSK_P2P_ERR_TIMEOUT) return;     // (k_p2p_wait gave up)
    p2p_unpack<VT>(val, val_evid, both, recv_iid, nrecv, mine, pl, world, tag, selftest, err);
    if (pw.w) p2p_reduce_slice<VT>(mine, nrecv, pl, pw, world, me, tag, selftest);
}
static __global__ void k_p2p_raise(P2PPlan pl, int kind, int world, int me, unsigned int peer_mask, unsigned int tag,
                                   const unsigned long long *tag_base) {
    if (tag_base) tag += (unsigned int)tag_base[1];
    p2p_raise(pl, kind, world, me, peer_mask, tag);
}
static __global__ void k_p2p_wait(const void *mine, int kind, int world, unsigned int peer_mask, unsigned int tag,
                                  const unsigned long long *tag_base, unsigned int *err, unsigned long long timeout_ticks) {
    if (tag_base) tag += (unsigned int)tag_base[1];
    (void)p2p_wait(mine, kind, world, peer_mask, tag, err, timeout_ticks);
}

// Partial-factor values (salt/src/messages.py:1333-1349 compute_pf_values): a reader shard that holds
// a factor OR / AND / ISTRUE with several members owned by THIS shard reads ONE aggregate of them instead of every
// member -- the factor's value over those members: "some member is 1" (OR) or "no member is 0" (AND, ISTRUE), stored as
// 1 / 0 in an extra value slot behind the internal ids, from where the exchange pushes it like any boundary value
// (the reader holds it as a boolean ghost variable in the members' place: f(a.., OR(b..)) = f(a.., b..) exactly).
template <typename VT>
__global__ __launch_bounds__(NSK_BLOCK) void k_pf_compute(VT *val, VT *val_evid, int both, const uint8_t *pf_op, const int32_t *pf_off,
                                                          const int32_t *pf_mem, int npf, long long base) {
    const int j = (int)(blockIdx.x * NSK_BLOCK + threadIdx.x);
    if (j >= npf) return;
    const bool is_or = pf_op[j] == 0;
    bool a = !is_or, b = !is_or;
    for (int k = pf_off[j]; k < pf_off[j + 1]; k++) {
        const int id = pf_mem[k];
        const int x = (int)val[id];
        a = is_or ? (a || x == 1) : (a && x != 0);
        if (both) { const int y = (int)val_evid[id]; b = is_or ? (b || y == 1) : (b && y != 0); }
    }
    val[base + j] = (VT)(a ? 1 : 0);
    if (both) val_evid[base + j] = (VT)(b ? 1 : 0);
}

// Set-up self-test of the FUSED exchange's memory protocol (nsk_kernels_gibbs.h TabP2P): what the border tiles of
// the table launches do, in isolation -- system-coherent stores of a pattern into the readers' receive blocks, wait
// for their acknowledgement, flag; then relaxed polls of the peers' flags and system-coherent loads of this rank's
// receive block, compared with what the peers must have written.  No acquire / release fence anywhere: if this
// device pair needs one for a peer's writes to become visible, the test fails and the ranks keep the exchange
// kernels.  One workgroup.
template <typename VT>
__global__ __launch_bounds__(NSK_BLOCK) void k_p2p_fused_selftest(long long nsend, long long nrecv, P2PPlan pl, const void *mine, int world, int me,
                                                                  unsigned int peer_mask, unsigned int tag, unsigned int *err,
                                                                  unsigned long long timeout_ticks, int part) {
    const size_t par = tag & 1u, roff = nsk_p2p_recv_off(world);
    if (part != 2) {
        for (long long k = threadIdx.x; k < nsend; k += NSK_BLOCK) {
            int q = 0;
            while (q + 1 < world && (unsigned long long)k >= pl.soff[q + 1]) q++;
            const unsigned long long kl = (unsigned long long)k - pl.soff[q];
            const nsk_rsrc rp = nsk_make_rsrc((VT *)((char *)pl.base[q] + roff) + par * 2 * (size_t)pl.dtotal[q]);
            nsk_buf_st_sys<VT>(rp, (uint32_t)(pl.dbase[q] + kl), (int)p2p_pattern<VT>(me, kl, tag, 0));
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);                   // vmcnt(0): the stores have been acknowledged
        __syncthreads();
        if (threadIdx.x < (unsigned)world && ((peer_mask >> threadIdx.x) & 1u))
            __hip_atomic_store((unsigned int *)pl.base[threadIdx.x] + (par * 2 + 0) * (size_t)world + (size_t)me, tag,
                               __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (part == 1) return;
    __shared__ int ok;
    if (threadIdx.x == 0) {
        ok = 1;
        const unsigned long long t0 = wall_clock64();
        const unsigned int *flags = (const unsigned int *)mine + (par * 2 + 0) * (size_t)world;
        for (int q = 0; q < world && ok; q++) {
            if (!((peer_mask >> q) & 1u)) continue;
            while (__hip_atomic_load(flags + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != tag) {
                if (wall_clock64() - t0 > timeout_ticks) { ok = 0; break; }
                __builtin_amdgcn_s_sleep(8);
            }
        }
        if (!ok) (void)__hip_atomic_fetch_or(err, NSK_P2P_ERR_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!ok) return;
    const nsk_rsrc rg = nsk_make_rsrc((const VT *)((const char *)mine + roff) + par * 2 * (size_t)nrecv);
    for (long long j = threadIdx.x; j < nrecv; j += NSK_BLOCK) {
        int q = 0;
        while (q + 1 < world && (unsigned long long)j >= pl.roff[q + 1]) q++;
        const unsigned long long jl = (unsigned long long)j - pl.roff[q];
        const uint32_t got = nsk_buf_ld_sys<VT>(rg, (uint32_t)j, 0u);
        if (got != ((uint32_t)(int)p2p_pattern<VT>(q, jl, tag, 0) & 0xFFu))
            (void)__hip_atomic_fetch_or(err, NSK_P2P_ERR_PAYLOAD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// the closing half of a learning exchange: wait for the owners' merged slices, then w = w_start = merged
// (the same vector on every rank)
template <typename VT>
__global__ __launch_bounds__(NSK_BLOCK) void k_p2p_gather_w(double *w, double *w_start, int nw, const void *mine, long long nrecv,
                                                            int world, unsigned int peer_mask, unsigned int tag, unsigned int *err,
                                                            unsigned long long timeout_ticks, int selftest, int waited) {
    // waited: a k_p2p_wait in front of this launch has seen the owners' flags (large tables: as many blocks as the
    // table wants, none of them polling)
    if (waited ? (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & NSK_P2P_ERR_TIMEOUT) != 0u
               : !p2p_wait(mine, 1, world, peer_mask, tag, err, timeout_ticks)) return;
    const double *gb = (const double *)((const char *)mine + nsk_p2p_gbuf_off(world, (size_t)nrecv, sizeof(VT), (size_t)nw)) +
                       (size_t)(tag & 1u) * (size_t)nw;
    for (int i = (int)(blockIdx.x * NSK_BLOCK + threadIdx.x); i < nw; i += (int)(gridDim.x * NSK_BLOCK)) {
        const double x = __builtin_nontemporal_load(gb + i);
        if (selftest) {
            double t = p2p_pattern_dw(0, i, tag);
            for (int r = 1; r < world; r++) t += p2p_pattern_dw(r, i, tag);
            if (x != 0.0 + t) (void)__hip_atomic_fetch_or(err, NSK_P2P_ERR_PAYLOAD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;
        }
        w[i] = x;
        w_start[i] = x;
    }
}

// counters of captured sweep sequences (hipGraph): [0] sweep index, [1] peer-to-peer exchange tag
// [2] the Philox key (seed), [3] the shard tag: a replay after nsk_set_seed / nsk_set_rng_tag draws from the new stream
static __global__ void k_graph_counters(unsigned long long *c, unsigned long long sweep, unsigned long long tag, int set,
                                        unsigned long long seed, unsigned long long rng_tag) {
    if (set) { c[0] = sweep; c[1] = tag; c[2] = seed; c[3] = rng_tag; }
    else { c[0] += sweep; c[1] += tag; }
}

// Self-test of the XCD-private accumulators (nsk_api.hip, once per device): every thread adds 1 to the
// slot of the XCD it runs on (HW_REG_XCC_ID) with a workgroup-scope atomic -- the add executes in that
// XCD's L2 -- and marks the id it saw.  The slots must add up to the number of threads: if two dies
// with incoherent L2s reported one id, or the adds did not reach memory at the kernel boundary, counts
// would be lost.  out[0..15]: slots, out[16]: mask of ids seen.
static __global__ void k_xcd_selftest(unsigned int *out) {
    const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & 15u;
    (void)__hip_atomic_fetch_add(&out[xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (threadIdx.x == 0) (void)__hip_atomic_fetch_or(&out[16], 1u << xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// weight merge of the partitioned learning sweep: delta = w - start ... w = start + sum(delta)
static __global__ __launch_bounds__(NSK_BLOCK) void k_weight_delta(const double *w, const double *start, double *delta, int n) {
    const int i = (int)(blockIdx.x * NSK_BLOCK + threadIdx.x);
    if (i < n) delta[i] = w[i] - start[i];
}
static __global__ __launch_bounds__(NSK_BLOCK) void k_weight_merge(double *w, const double *start, const double *delta, int n) {
    const int i = (int)(blockIdx.x * NSK_BLOCK + threadIdx.x);
    if (i < n) w[i] = start[i] + delta[i];
}

static __global__ void k_selftest_exp(const double *x, double *y, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = nsk_exp(x[i]);
}

static __global__ void k_selftest_philox(uint32_t k0, uint32_t k1, uint32_t stream, uint32_t s0, uint32_t s1,
                                  long long n, uint32_t *out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32x4 r = philox4x32(k0, k1, (uint32_t)i, stream, s0, s1);
    out[4 * i] = r.x; out[4 * i + 1] = r.y; out[4 * i + 2] = r.z; out[4 * i + 3] = r.w;
}

// position-indexed tally deltas of the fast path -> int64 master copy at cstart[vid]
// packed tally (k_gibbs_seg_tabw, burnin == 2) back to the two arrays: cnt_pos += byte >> 1, value = byte & 1, sixteen
// positions per thread
static __global__ __launch_bounds__(NSK_BLOCK) void k_unpack_tally(uint4 *val, uint4 *cnt_pos, long long n16) {
    const long long i = (long long)blockIdx.x * NSK_BLOCK + threadIdx.x;
    if (i >= n16) return;
    uint4 v = val[i], c = cnt_pos[i];
    c.x += (v.x >> 1) & 0x7F7F7F7Fu; c.y += (v.y >> 1) & 0x7F7F7F7Fu; c.z += (v.z >> 1) & 0x7F7F7F7Fu; c.w += (v.w >> 1) & 0x7F7F7F7Fu;
    v.x &= 0x01010101u; v.y &= 0x01010101u; v.z &= 0x01010101u; v.w &= 0x01010101u;
    val[i] = v;
    cnt_pos[i] = c;
}
static __global__ __launch_bounds__(NSK_BLOCK) void k_fold_counts_pos(uint8_t *cnt_pos, const int32_t *p_cnt,
                                                               const int32_t *p_vid, long long *total, int npos) {
    const int i = (int)(blockIdx.x * NSK_BLOCK + threadIdx.x);
    if (i >= npos) return;
    const int d = cnt_pos[i];
    if (d) {
        if (p_vid[i] >= 0) total[p_cnt[i]] += (long long)d;      // (padding lanes of the table kernels tally too)
        cnt_pos[i] = 0;
    }
}

// ---- sequential validation scan: one lane walks variable ids in order with MT19937 -----------
template <typename VT>
__global__ void k_seq_gibbs(DevGraph<VT> g, const int32_t *v_pos, MTState *np_rng, int nsweeps,
                            int sample_evidence, int burnin) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    for (int s = 0; s < nsweeps; s++) {
        for (int v = 0; v < g.nvar; v++) {
            const int p = v_pos[v];
            if (p < 0) continue;
            const uint32_t info = g.p_info[p];
            if (!(NSK_INFO_EV(info) == 0 || sample_evidence)) continue;
            // the reference fills Z first and draws its uniform afterwards; draw_sample only needs
            // u at the very end, and nothing else consumes the stream in between
            const double u = mt_res53(np_rng);
            const int nv = draw_sample(g, p, info, g.p_slot[p], g.val, u);
            g.val[p] = (VT)nv;
            if (!burnin) {
                const int base = g.p_cnt[p];
                if (NSK_INFO_CARD(info) == 2) g.cnt[base] += nv;
                else g.cnt[base + nv] += 1;
            }
        }
    }
}

template <typename VT>
__global__ void k_seq_learn(DevGraph<VT> g, const int32_t *v_pos, MTState *np_rng, MTState *py_rng,
                            int nsweeps, double step, double decay, int regularization,
                            double reg_param, double truncation, int learn_non_evidence) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    for (int s = 0; s < nsweeps; s++) {
        for (int v = 0; v < g.nvar; v++) {
            const int p = v_pos[v];
            if (p < 0) continue;
            const uint32_t info = g.p_info[p];
            const int ev = NSK_INFO_EV(info);
            const int slot0 = g.p_slot[p];
            int evidence;
            if (ev != 1) evidence = draw_sample(g, p, info, slot0, g.val_evid, mt_res53(np_rng));
            else evidence = (int)g.p_init[p];
            g.val_evid[p] = (VT)evidence;
            const int proposal = draw_sample(g, p, info, slot0, g.val, mt_res53(np_rng));
            g.val[p] = (VT)proposal;
            if (!learn_non_evidence && ev != 1) continue;
            const int st = NSK_INFO_DT1(info);
            int a = g.slot_off[slot0 + st * evidence], ae = g.slot_off[slot0 + st * evidence + 1];
            int b = 0, be = 0;
            if (st && evidence != proposal) {
                b = g.slot_off[slot0 + proposal];
                be = g.slot_off[slot0 + proposal + 1];
            }
            bool truncate = false;
            if (regularization == 1) truncate = mt_res53(py_rng) < 1.0 / truncation;
            while (a < ae || b < be) {          // sorted, de-duplicated union == learning.py:76-98
                const int fa = a < ae ? g.fidx[a] : 0x7fffffff;
                const int fb = b < be ? g.fidx[b] : 0x7fffffff;
                const int fid = fa < fb ? fa : fb;
                if (fa == fid) a++;
                if (fb == fid) b++;
                const uint4 rec = g.f_rec[fid];
                const int wid = (int)rec.z;
                if (g.w_fixed[wid]) continue;
                const double p0 = eval_factor(g, rec, g.m_rec, p, evidence, g.val_evid);
                const double p1 = eval_factor(g, rec, g.m_rec, p, proposal, g.val);
                const double gradient = (p1 - p0) * g.f_feat[fid];
                double w = g.w[wid];
                if (regularization == 2) {
                    w *= (1.0 / (1.0 + reg_param * step));
                    w -= step * gradient;
                } else if (regularization == 1) {
                    w -= step * gradient;
                    if (truncate) {
                        const double l1delta = reg_param * step * truncation;
                        w = (w > 0) ? fmax(0.0, w - l1delta) : fmin(0.0, w + l1delta);
                    }
                } else {
                    w -= step * gradient;
                }
                g.w[wid] = w;
            }
        }
        step *= decay;
    }
}

}  // namespace nsk
