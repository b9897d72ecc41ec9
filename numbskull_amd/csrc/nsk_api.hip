// nsk_api.hip -- sweep kernels (gfx950) and the C-ABI entry points of include/numbskull_amd.h.
//
// Replaces the callee side of the reference's three run_pool(...) call sites
// (numbskull/factorgraph.py:141,163,202): gibbsthread (inference.py:10-33) and
// learnthread/sample_and_sgd (learning.py:12-125).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>      // types only: the library is bound at run time with dlopen

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/numbskull_amd.h"
#include "nsk_compile.h"
#include "nsk_device.h"

#include "nsk_internal.h"
#include "nsk_kernels_learn.h"
#include "nsk_kernels_misc.h"

using namespace nsk;

// =============================================================================================
// host side
// =============================================================================================
static thread_local std::string g_err;

namespace nsk {
int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
void set_error(const std::string &m) { g_err = m; }
}

// RCCL entry points, bound at run time (nsk_comm_init)
struct RcclApi {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
static RcclApi g_rccl;





template <typename T>
static int dev_alloc(nsk_graph *g, T **ptr, size_t n) {
    size_t bytes = (n ? n : 1) * sizeof(T);
    void *p = nullptr;
    // (diagnostic, NSK_DIAG=1 NSK_ALLOC_ALIGN=1: arrays of a megabyte or more start on a 2 MB boundary whatever the
    // allocator does -- where the arrays of the table kernels lie moves their launch time by 3 %)
    static const bool align2m = nsk::diag_env("NSK_ALLOC_ALIGN") != nullptr;
    const size_t big = (size_t)1 << 20, two = (size_t)2 << 20;
    const bool al = align2m && bytes >= big;
    hipError_t e = hipMalloc(&p, al ? bytes + two : bytes);
    if (e != hipSuccess) return fail(NSK_E_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    g->allocs.push_back(p);
    g->device_bytes += (int64_t)bytes;
    if (al) {
        void *raw = p;
        p = (void *)(((uintptr_t)p + two - 1) / two * two);
        if (p != raw) g->alloc_alias.push_back({p, raw});       // (dev_free must hand hipFree the allocation, not the aligned pointer)
    }
    *ptr = (T *)p;
    return NSK_OK;
}

// frees an array dev_alloc handed out before the handle is destroyed (arrays that are re-allocated: nsk_pf_setup)
static void dev_free(nsk_graph *g, void *p) {
    if (!p) return;
    void *raw = p;
    for (size_t i = 0; i < g->alloc_alias.size(); i++)
        if (g->alloc_alias[i].first == p) { raw = g->alloc_alias[i].second; g->alloc_alias.erase(g->alloc_alias.begin() + (long)i); break; }
    g->allocs.erase(std::remove(g->allocs.begin(), g->allocs.end(), raw), g->allocs.end());
    (void)hipFree(raw);
}

template <typename T>
static int dev_upload(nsk_graph *g, T **ptr, const std::vector<T> &h) {
    int rc = dev_alloc(g, ptr, h.size());
    if (rc) return rc;
    if (!h.empty()) HIPCHECK(hipMemcpyAsync(*ptr, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, g->stream));
    return NSK_OK;
}

// May this device keep XCD-private accumulators (workgroup-scope atomics in the issuing XCD's L2,
// private copies picked by HW_REG_XCC_ID)?  gfx942 / gfx950 by architecture name AND a self-test, run
// once per device and process: 2048 x 256 threads add to the slot of their XCD (k_xcd_selftest); the
// slots, read back after the kernel boundary, must add up to the number of threads, with ids < 8.
static bool xcd_private_ok(nsk_graph *g) {
    static std::mutex mu;
    static std::map<int, bool> verdict;
    std::lock_guard<std::mutex> lk(mu);
    auto it = verdict.find(g->device);
    if (it != verdict.end()) return it->second;
    bool ok = false;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, g->device) == hipSuccess &&
        (strstr(prop.gcnArchName, "gfx950") || strstr(prop.gcnArchName, "gfx942"))) {
        unsigned int *d = nullptr, h[17];
        const unsigned int nblocks = 2048, nthreads = 256;
        if (hipMalloc((void **)&d, sizeof(h)) == hipSuccess) {
            bool ran = hipMemsetAsync(d, 0, sizeof(h), g->stream) == hipSuccess;
            if (ran) {
                k_xcd_selftest<<<dim3(nblocks), dim3(nthreads), 0, g->stream>>>(d);
                ran = hipGetLastError() == hipSuccess &&
                      hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, g->stream) == hipSuccess &&
                      hipStreamSynchronize(g->stream) == hipSuccess;
            }
            if (ran) {
                unsigned long long sum = 0;
                for (int i = 0; i < 16; i++) sum += h[i];
                ok = sum == (unsigned long long)nblocks * nthreads && h[16] != 0u && (h[16] >> NSK_XCDS) == 0u;
            }
            (void)hipFree(d);
        }
        if (!ok && getenv("NSK_VERBOSE")) fprintf(stderr, "[nsk] XCD-private accumulators: self-test failed, one shared copy\n");
    }
    verdict[g->device] = ok;
    return ok;
}

// narrow int32 host values to the device value type (int8 / int32) and upload
static int upload_values(nsk_graph *g, void *dst, const int32_t *src, size_t n) {
    if (g->c.vbytes == 4) {
        if (n) HIPCHECK(hipMemcpyAsync(dst, src, n * 4, hipMemcpyHostToDevice, g->stream));
        HIPCHECK(hipStreamSynchronize(g->stream));
        return NSK_OK;
    }
    std::vector<int8_t> tmp(n);
    for (size_t i = 0; i < n; i++) tmp[i] = (int8_t)src[i];
    if (n) HIPCHECK(hipMemcpyAsync(dst, tmp.data(), n, hipMemcpyHostToDevice, g->stream));
    HIPCHECK(hipStreamSynchronize(g->stream));
    return NSK_OK;
}

static void mt_seed_numpy(MTState &s, uint32_t seed) {      // np.random.seed(int): init_genrand
    s.mt[0] = seed;
    for (int i = 1; i < 624; i++) s.mt[i] = 1812433253u * (s.mt[i - 1] ^ (s.mt[i - 1] >> 30)) + (uint32_t)i;
    s.idx = 624;
}

static void mt_seed_python(MTState &s, uint64_t seed) {     // random.seed(int): init_by_array
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    int keylen = key[1] ? 2 : 1;
    mt_seed_numpy(s, 19650218u);
    int i = 1, j = 0;
    for (int k = 624; k; k--) {
        s.mt[i] = (s.mt[i] ^ ((s.mt[i - 1] ^ (s.mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        i++; j++;
        if (i >= 624) { s.mt[0] = s.mt[623]; i = 1; }
        if (j >= keylen) j = 0;
    }
    for (int k = 623; k; k--) {
        s.mt[i] = (s.mt[i] ^ ((s.mt[i - 1] ^ (s.mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
        i++;
        if (i >= 624) { s.mt[0] = s.mt[623]; i = 1; }
    }
    s.mt[0] = 0x80000000u;
    s.idx = 624;
}


extern "C" {

const char *nsk_last_error(void) { return g_err.c_str(); }
const char *nsk_version(void) { return "numbskull_amd 0.1.1 (gfx950)"; }

int nsk_device_count(int *count) {
    if (!count) return fail(NSK_E_INVALID, "null count");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(NSK_E_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
    *count = n;
    return NSK_OK;
}

int nsk_graph_destroy(nsk_graph *g) {
    if (!g) return NSK_OK;
    (void)hipSetDevice(g->device);
    if (g->stream) (void)hipStreamSynchronize(g->stream);
    for (int q = 0; q < 16; q++)           // peers' allocations mapped with hipIpc
        if (g->p2p_peer_ipc[q] && g->p2p_peer_base[q]) (void)hipIpcCloseMemHandle(g->p2p_peer_base[q]);
    for (void *p : g->allocs) (void)hipFree(p);
    for (int k = 0; k < 2; k++) if (g->xfer_host[k]) (void)hipHostFree(g->xfer_host[k]);
    if (g->cnt_host) (void)hipHostFree(g->cnt_host);
    if (g->sweep_graph) (void)hipGraphExecDestroy(g->sweep_graph);
    if (g->sweep_graph_big) (void)hipGraphExecDestroy(g->sweep_graph_big);
    if (g->ev0) (void)hipEventDestroy(g->ev0);
    if (g->ev1) (void)hipEventDestroy(g->ev1);
    if (g->rccl_comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy((ncclComm_t)g->rccl_comm);
    if (g->own_stream && g->stream) (void)hipStreamDestroy(g->stream);
    for (int i = 0; i < 3; i++) {
        if (g->side[i]) (void)hipStreamDestroy(g->side[i]);
        if (g->ev_join[i]) (void)hipEventDestroy(g->ev_join[i]);
    }
    if (g->ev_fork) (void)hipEventDestroy(g->ev_fork);
    delete g;
    return NSK_OK;
}

// upload the generic-path arrays (once)
int nsk_ensure_generic(nsk_graph *g) {
    if (g->generic_uploaded) return NSK_OK;
    Compiled &c = g->c;
    int rc;
    HIPCHECK(hipSetDevice(g->device));
#define UP(name) do { rc = dev_upload(g, &g->name, c.name); if (rc) return rc; } while (0)
    UP(p_slot); UP(slot_off); UP(fidx); UP(gstream); UP(gs_off);
    UP(f_rec); UP(f_feat); UP(m_rec); UP(v_pos);
#undef UP
    rc = dev_upload(g, &g->v_card, c.v_card_i); if (rc) return rc;
    if (c.literal_heads) { rc = dev_upload(g, &g->iid_of_vid, c.iid); if (rc) return rc; }
    HIPCHECK(hipStreamSynchronize(g->stream));
    g->generic_uploaded = true;
    return NSK_OK;
}

static int create_impl(const nsk_graph_desc *desc, nsk_graph *g) {
    std::string err;
    const auto t_compile = std::chrono::steady_clock::now();
    int rc = compile_graph(desc, g->c, err);
    if (rc) return fail(rc, err);
    g->compile_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_compile).count();
    g->values_regular = g->chain_regular[0] = g->chain_regular[1] = g->c.values_regular;
    g->rng_tag = (uint32_t)g->c.own_begin;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(NSK_E_DEVICE, "no HIP device available: the Gibbs sweep runs on the GPU only "
                                  "(there is no CPU fallback)");
    if (desc->device < 0 || desc->device >= ndev) return fail(NSK_E_INVALID, "device ordinal out of range");
    g->device = desc->device;
    HIPCHECK(hipSetDevice(g->device));
    HIPCHECK(hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking));
    g->own_stream = true;
    for (int i = 0; i < 3; i++) {
        HIPCHECK(hipStreamCreateWithFlags(&g->side[i], hipStreamNonBlocking));
        HIPCHECK(hipEventCreateWithFlags(&g->ev_join[i], hipEventDisableTiming));
    }
    HIPCHECK(hipEventCreateWithFlags(&g->ev_fork, hipEventDisableTiming));
    HIPCHECK(hipEventCreate(&g->ev0));
    HIPCHECK(hipEventCreate(&g->ev1));
    Compiled &c = g->c;
#define UP(name) do { rc = dev_upload(g, &g->name, c.name); if (rc) return rc; } while (0)
    if (const char *padn = nsk::diag_env("NSK_ALLOC_PAD"))          // (diagnostic: n tiny allocations in front of the arrays)
        for (int i = 0; i < atoi(padn); i++) { uint32_t *dummy = nullptr; rc = dev_alloc(g, &dummy, 1); if (rc) return rc; }
    UP(p_vid); UP(p_info); UP(p_cnt);
    // the CSR-style arrays serve the generic kernels, the hub waves, the per-lane-header tiles and
    // the sequential validation scan only: a graph that lives entirely in tiles (the Ising grids:
    // 1.4 GB of them at 10M variables) uploads them when the sequential scan is first selected
    {
        bool generic_needed = c.phase_dyn_base.size() > 0 && c.phase_dyn_base.back() > 0;
        for (size_t k = 0; k + 1 < c.phase_start.size(); k++)
            if (c.phase_end[k] > c.phase_fast_end[k]) generic_needed = true;
        if (generic_needed || nsk::diag_env("NSK_EAGER_GENERIC")) { rc = nsk_ensure_generic(g); if (rc) return rc; }
    }
    UP(w_fixed); UP(logtab); UP(adj); UP(seg_aff); UP(seg_wide); UP(wide_exc); UP(hub_desc); UP(hub_adj); UP(ep_desc); UP(ep_adj); UP(ep_wrow); UP(ep_kstat); UP(bighub_pos); UP(tiles); UP(tile_hdr); UP(dyn_tiles); UP(rest_tiles); UP(learn_rest_tiles); UP(tile_wrow);
#undef UP
    // (value windows, -DNSK_EP_WIN builds only: no allocation otherwise -- the default build's sequence of device
    // allocations is round 4's, address for address)
    if (!c.ep_win.empty()) {
        rc = dev_upload(g, &g->ep_win, c.ep_win); if (rc) return rc;
        rc = dev_upload(g, &g->ep_win_off, c.ep_win_off); if (rc) return rc;
    }
    rc = dev_upload(g, &g->w, c.w_init); if (rc) return rc;
    if (c.ndirect > 0) {
        rc = dev_upload(g, &g->w_direct, c.w_direct); if (rc) return rc;
        rc = dev_upload(g, &g->multi_wids, c.multi_wids); if (rc) return rc;
    }
    const size_t nvar = (size_t)c.nvar, npos = (size_t)c.npos, vb = (size_t)c.vbytes, nid = (size_t)c.nid;
    uint8_t *tmp = nullptr;
    rc = dev_alloc(g, &tmp, npos * vb); if (rc) return rc; g->p_init = tmp;
    // (+ 16 bytes with value windows: they are copied in whole 16-byte chunks)
    const size_t vpad = c.ep_win.empty() ? 0 : 16;
    rc = dev_alloc(g, &tmp, nid * vb + vpad + 16); if (rc) return rc; g->val = tmp;      // (+ 16: k_unpack_tally walks 16-byte chunks)
    rc = dev_alloc(g, &tmp, nid * vb + vpad + 16); if (rc) return rc; g->val_evid = tmp;
    rc = upload_values(g, g->p_init, c.p_init.data(), npos); if (rc) return rc;
    {   // values live at internal ids (padding positions hold 0 and are never read as a variable)
        std::vector<int32_t> init_i(nid, 0);
        for (size_t v = 0; v < nvar; v++) init_i[c.iid[v]] = c.v_init[v];
        rc = upload_values(g, g->val, init_i.data(), nid); if (rc) return rc;
        rc = upload_values(g, g->val_evid, init_i.data(), nid); if (rc) return rc;
    }
    rc = dev_alloc(g, &g->cnt, (size_t)c.ncount); if (rc) return rc;
    rc = dev_alloc(g, &g->cnt_total, (size_t)c.ncount); if (rc) return rc;
    rc = dev_alloc(g, &g->cnt_pos, (size_t)c.npos + 16); if (rc) return rc;
    rc = dev_alloc(g, &g->prog_w, 2 * c.tile_hdr.size()); if (rc) return rc;
    rc = dev_alloc(g, &g->adj_wt, (size_t)c.nwrows * 64); if (rc) return rc;
    rc = dev_alloc(g, &g->ztab, (size_t)c.nztab); if (rc) return rc;
    rc = dev_alloc(g, &g->ep_wt, (size_t)(c.ep_wrow.empty() ? 0 : c.ep_wrow.back()) * 64); if (rc) return rc;
    rc = dev_alloc(g, &g->sink, 1024); if (rc) return rc;
    if (getenv("NSK_VERBOSE"))
        fprintf(stderr, "[nsk] device arrays: val %p val_evid %p cnt_pos %p seg_aff %p adj %p ztab %p p_init %p (mod 2 MB: %zx %zx %zx %zx)\n",
                g->val, g->val_evid, (void *)g->cnt_pos, (void *)g->seg_aff, (void *)g->adj, (void *)g->ztab, g->p_init,
                (size_t)((uintptr_t)g->val & 0x1FFFFF), (size_t)((uintptr_t)g->cnt_pos & 0x1FFFFF), (size_t)((uintptr_t)g->seg_aff & 0x1FFFFF),
                (size_t)((uintptr_t)g->val_evid & 0x1FFFFF));
    {
        std::vector<ZProgDev> zp(c.zprogs.size());
        for (size_t i = 0; i < zp.size(); i++) zp[i] = {c.zprogs[i].prog, c.zprogs[i].nslots, c.zprogs[i].off, 0u};
        rc = dev_upload(g, &g->zprogs, zp); if (rc) return rc;
    }
    g->smallw = c.nweight > 0 && c.nweight <= NSK_SMALLW;
    if (g->smallw) {
        const size_t cells = (size_t)NSK_LEARN_BINS * (size_t)c.nweight;     // binned partial sums
        rc = dev_alloc(g, &g->part_G, cells); if (rc) return rc;
        rc = dev_alloc(g, &g->part_K, cells); if (rc) return rc;
        rc = dev_alloc(g, &g->part_T, cells); if (rc) return rc;
        HIPCHECK(hipMemsetAsync(g->part_G, 0, cells * sizeof(long long), g->stream));
        HIPCHECK(hipMemsetAsync(g->part_K, 0, cells * sizeof(uint32_t), g->stream));
        HIPCHECK(hipMemsetAsync(g->part_T, 0, cells * sizeof(uint32_t), g->stream));
    }
    HIPCHECK(hipMemsetAsync(g->cnt_pos, 0, (size_t)c.npos + 16, g->stream));
    // global learning accumulators: one private copy per XCD (nsk_device.h sink_add); graphs with few
    // weights accumulate in LDS and never touch them.  A private copy pays while it stays in its XCD's L2
    // (4 MB): up to 2^18 weights (2 MB of sums).  Beyond, the adds miss the L2 either way and eight copies only
    // multiply the update launch's reads -- 50M LR graph, 10^6 weights (tools/sessions/history/r4_s22.sh): eight copies
    // 5.22e9 updates/s, one copy with agent-scope adds 5.47e9; 5M LR graph, 10^5 weights: copies 262 us per
    // class against 300.
    g->acc_copies = (!g->smallw && c.nweight <= (1 << 18)) ? NSK_XCDS : 1;
    {   // the XCD-private copies rely on global atomics executing in the issuing XCD's own L2 and on
        // HW_REG_XCC_ID (nsk_device.h sink_add): true on gfx942 / gfx950 -- and checked once per device by
        // xcd_private_ok -- so any other architecture, a failed check (or NSK_DIAG=1 NSK_ONE_ACC=1) keeps
        // ONE copy updated with agent-scope atomics
        const bool known = xcd_private_ok(g);
        if (!known || nsk::diag_env("NSK_ONE_ACC")) g->acc_copies = 1;
        g->bins_xcd = (known && !nsk::diag_env("NSK_ONE_ACC")) ? 1 : 0;
    }
    const size_t nacc = (size_t)g->acc_copies * (size_t)(c.nweight ? c.nweight : 1);
    rc = dev_alloc(g, &g->G, nacc); if (rc) return rc;
    rc = dev_alloc(g, &g->K, nacc); if (rc) return rc;
    rc = dev_alloc(g, &g->T, nacc); if (rc) return rc;
    rc = dev_alloc(g, &g->clip_count, 1); if (rc) return rc;
    rc = dev_alloc(g, &g->d_counters, 4); if (rc) return rc;
    HIPCHECK(hipMemsetAsync(g->clip_count, 0, sizeof(unsigned int), g->stream));
    rc = dev_alloc(g, &g->mt_np, 1); if (rc) return rc;
    rc = dev_alloc(g, &g->mt_py, 1); if (rc) return rc;
    HIPCHECK(hipMemsetAsync(g->cnt, 0, (c.ncount ? c.ncount : 1) * sizeof(int32_t), g->stream));
    HIPCHECK(hipMemsetAsync(g->cnt_total, 0, (c.ncount ? c.ncount : 1) * sizeof(long long), g->stream));
    HIPCHECK(hipMemsetAsync(g->G, 0, nacc * sizeof(long long), g->stream));
    HIPCHECK(hipMemsetAsync(g->K, 0, nacc * sizeof(uint32_t), g->stream));
    HIPCHECK(hipMemsetAsync(g->T, 0, nacc * sizeof(uint32_t), g->stream));
    HIPCHECK(hipStreamSynchronize(g->stream));
    return nsk_set_seed(g, 0, 0);
}

int nsk_graph_create(const nsk_graph_desc *desc, nsk_graph **out) {
    if (!desc || !out) return fail(NSK_E_INVALID, "null argument");
    *out = nullptr;
    nsk_graph *g = new nsk_graph();
    int rc = create_impl(desc, g);
    if (rc) {
        std::string keep = g_err;
        nsk_graph_destroy(g);
        g_err = keep;
        return rc;
    }
    *out = g;
    return NSK_OK;
}

int nsk_set_seed(nsk_graph *g, uint64_t seed, uint64_t sweep0) {
    if (!g) return fail(NSK_E_INVALID, "null graph");
    HIPCHECK(hipSetDevice(g->device));
    g->seed = seed;
    g->sweep = sweep0;
    MTState a, b;
    mt_seed_numpy(a, (uint32_t)seed);
    mt_seed_python(b, seed);
    HIPCHECK(hipMemcpyAsync(g->mt_np, &a, sizeof(MTState), hipMemcpyHostToDevice, g->stream));
    HIPCHECK(hipMemcpyAsync(g->mt_py, &b, sizeof(MTState), hipMemcpyHostToDevice, g->stream));
    HIPCHECK(hipStreamSynchronize(g->stream));
    return NSK_OK;
}

int nsk_set_rng_tag(nsk_graph *g, uint32_t tag) {
    if (!g) return fail(NSK_E_INVALID, "null graph");
    g->rng_tag = tag;
    return NSK_OK;
}

int nsk_set_learn_cap(nsk_graph *g, double cap) {
    if (!g) return fail(NSK_E_INVALID, "null graph");
    if (cap != cap) return fail(NSK_E_INVALID, "cap is NaN");
    g->learn_cap = cap;
    return NSK_OK;
}

int nsk_set_learn_lag(nsk_graph *g, int lag) {
    if (!g) return fail(NSK_E_INVALID, "null graph");
    g->learn_lag = lag != 0;
    return NSK_OK;
}

int nsk_set_scan(nsk_graph *g, int scan) {
    if (!g) return fail(NSK_E_INVALID, "null graph");
    if (scan != NSK_SCAN_CHROMATIC && scan != NSK_SCAN_SEQUENTIAL) return fail(NSK_E_INVALID, "unknown scan order");
    if (scan == NSK_SCAN_SEQUENTIAL && (g->c.own_begin != 0 || g->c.own_end != g->c.nvar))
        return fail(NSK_E_INVALID, "sequential scan needs the whole graph on one handle");
    if (scan == NSK_SCAN_SEQUENTIAL) { int rc = nsk_ensure_generic(g); if (rc) return rc; }
    g->scan = scan;
    return NSK_OK;
}

int nsk_set_stream(nsk_graph *g, void *hip_stream) {
    if (!g) return fail(NSK_E_INVALID, "null graph");
    HIPCHECK(hipSetDevice(g->device));
    HIPCHECK(hipStreamSynchronize(g->stream));
    if (g->own_stream) { HIPCHECK(hipStreamDestroy(g->stream)); g->own_stream = false; }
    g->stream = (hipStream_t)hip_stream;
    return NSK_OK;
}

int nsk_synchronize(nsk_graph *g) {
    if (!g) return fail(NSK_E_INVALID, "null graph");
    HIPCHECK(hipSetDevice(g->device));
    { int frc = nsk_p2p_flush(g); if (frc) return frc; }
    HIPCHECK(hipStreamSynchronize(g->stream));
    return NSK_OK;
}

}  // extern "C"

// the fast path reads weights through prog_w: rebuild it whenever weights may have changed (start
// of every sweep call -- the host may have written the weight buffer -- and after every update)
void nsk_refresh_ztab(nsk_graph *g, int set, hipStream_t st) {
    if (!g->c.zprogs.empty())
        k_refresh_ztab<<<dim3((unsigned)g->c.zprogs.size()), dim3(NSK_BLOCK), 0, st ? st : g->stream>>>(
            g->zprogs, g->tile_hdr, set ? g->prog_w1 : g->prog_w, set ? g->ztab1 : g->ztab);
}

int nsk_ensure_lag_sets(nsk_graph *g) {
    if (g->G1) return NSK_OK;
    const Compiled &c = g->c;
    int rc;
    HIPCHECK(hipSetDevice(g->device));
    if ((rc = dev_alloc(g, &g->w1, (size_t)c.nweight))) return rc;
    if ((rc = dev_alloc(g, &g->prog_w1, 2 * c.tile_hdr.size()))) return rc;
    if ((rc = dev_alloc(g, &g->ztab1, (size_t)c.nztab))) return rc;
    if (g->smallw) {
        const size_t cells = (size_t)NSK_LEARN_BINS * (size_t)c.nweight;
        if ((rc = dev_alloc(g, &g->part_G1, cells))) return rc;
        if ((rc = dev_alloc(g, &g->part_K1, cells))) return rc;
        if ((rc = dev_alloc(g, &g->part_T1, cells))) return rc;
        HIPCHECK(hipMemsetAsync(g->part_G1, 0, cells * sizeof(long long), g->stream));
        HIPCHECK(hipMemsetAsync(g->part_K1, 0, cells * sizeof(uint32_t), g->stream));
        HIPCHECK(hipMemsetAsync(g->part_T1, 0, cells * sizeof(uint32_t), g->stream));
    }
    const size_t nacc = (size_t)g->acc_copies * (size_t)(c.nweight ? c.nweight : 1);
    if ((rc = dev_alloc(g, &g->G1, nacc))) return rc;
    if ((rc = dev_alloc(g, &g->K1, nacc))) return rc;
    if ((rc = dev_alloc(g, &g->T1, nacc))) return rc;
    HIPCHECK(hipMemsetAsync(g->G1, 0, nacc * sizeof(long long), g->stream));
    HIPCHECK(hipMemsetAsync(g->K1, 0, nacc * sizeof(uint32_t), g->stream));
    HIPCHECK(hipMemsetAsync(g->T1, 0, nacc * sizeof(uint32_t), g->stream));
    HIPCHECK(hipStreamSynchronize(g->stream));
    return NSK_OK;
}

void nsk_refresh_prog_weights(nsk_graph *g, bool force) {
    if (!force && !g->weights_dirty && !g->weights_exposed) return;
    g->weights_dirty = false;
    const int n = (int)g->c.tile_hdr.size();
    if (n > 0 && g->c.nfast > 0 && g->c.nweight > 0) {
        k_refresh_prog_weights<<<dim3((n + NSK_BLOCK - 1) / NSK_BLOCK), dim3(NSK_BLOCK), 0, g->stream>>>(
            g->tile_hdr, g->w, g->prog_w, n);
        nsk_refresh_ztab(g);
    }
    const int neg = (int)g->c.ep_wrow.size() - 1;
    if (neg > 0 && !g->adj_wt_skip)          // materialised weights of the entry-parallel groups (inference)
        k_refresh_ep_weights<<<dim3((unsigned)neg), dim3(NSK_BLOCK), 0, g->stream>>>(
            (const uint4 *)g->ep_desc, g->ep_adj, g->ep_wrow, g->w, g->ep_wt);
    const int nt = (int)(g->c.tiles.size() / 4) - 1;
    if (g->c.nwrows > 0 && nt > 0 && !g->adj_wt_skip)
        k_refresh_shape_weights<<<dim3((nt + 3) / 4), dim3(NSK_BLOCK), 0, g->stream>>>(
            (const uint4 *)g->tiles, (const uint4 *)g->adj, g->tile_hdr, g->tile_wrow, g->w, g->adj_wt, nt);
}

// the tally a packed-mode sweep sequence left inside the value bytes (nsk_gibbs.hip pack_tally) back into cnt_pos
int nsk_unpack_tally(nsk_graph *g) {
    if (g->packed_sweeps == 0) return NSK_OK;
    const long long n16 = ((long long)g->c.npos + 15) / 16;
    if (n16 > 0)
        k_unpack_tally<<<dim3((unsigned)((n16 + NSK_BLOCK - 1) / NSK_BLOCK)), dim3(NSK_BLOCK), 0, g->stream>>>((uint4 *)g->val, (uint4 *)g->cnt_pos, n16);
    g->packed_sweeps = 0;
    return NSK_OK;
}

int nsk_fold_position_tally(nsk_graph *g) {
    (void)nsk_unpack_tally(g);
    const int np = (int)g->c.npos;
    if (np > 0 && g->c.nfast > 0 && g->pos_tally_sweeps > 0)
        k_fold_counts_pos<<<dim3((np + NSK_BLOCK - 1) / NSK_BLOCK), dim3(NSK_BLOCK), 0, g->stream>>>(
            g->cnt_pos, g->p_cnt, g->p_vid, g->cnt_total, np);
    g->pos_tally_sweeps = 0;
    return NSK_OK;
}

extern "C" {

static int fold_counts(nsk_graph *g) {
    if (!g->cnt_dirty) return NSK_OK;
    const int n = (int)g->c.ncount;
    if (n > 0)
        k_fold_counts<<<dim3((n + NSK_BLOCK - 1) / NSK_BLOCK), dim3(NSK_BLOCK), 0, g->stream>>>(
            g->cnt, g->cnt_total, n);
    nsk_fold_position_tally(g);
    HIPCHECK(hipGetLastError());
    g->cnt_dirty = false;
    return NSK_OK;
}

}  // extern "C"



static int xfer_ensure(nsk_graph *g) {
    if (g->xfer_dev) return NSK_OK;
    const size_t bytes = (size_t)std::max<int64_t>(g->c.nvar, 1) * (size_t)g->c.vbytes;
    // (non-coherent = cacheable on the host: the host threads READ these buffers after a download, and reads of
    //  the default, uncached mapping ran at ~1 GB/s)
    for (int k = 0; k < 2; k++) HIPCHECK(hipHostMalloc(&g->xfer_host[k], bytes, hipHostMallocNonCoherent));
    uint8_t *t = nullptr;
    int rc = dev_alloc(g, &t, bytes);
    if (rc) return rc;
    g->xfer_dev = t;
    if (g->iid_of_vid) { g->xfer_iid = g->iid_of_vid; return NSK_OK; }
    return dev_upload(g, &g->xfer_iid, g->c.iid);
}

// validate the caller's int64 values and narrow them to the device value type, in the caller's order
// (host threads over index blocks); bad: 0 fine, 1 a value does not fit the value type; *regular: every
// value lies in [0, cardinality)
template <typename VT>
static void narrow_values(const nsk_graph *g, const int64_t *src, VT *dst, int *bad, bool *regular) {
    const int64_t lo = sizeof(VT) == 1 ? -128 : INT32_MIN, hi = sizeof(VT) == 1 ? 127 : INT32_MAX;
    const int32_t *card = g->c.v_card.data();
    std::vector<int> tbad((size_t)nsk::compile_threads(), 0), tirr((size_t)nsk::compile_threads(), 0);
    nsk::parallel_for(g->c.nvar, [&](int64_t b0, int64_t b1, int t) {
        int bd = 0, ir = 0;
        for (int64_t i = b0; i < b1; i++) {
            const int64_t x = src[i];
            bd |= (x < lo || x > hi) ? 1 : 0;
            ir |= (x < 0 || x >= (int64_t)card[i]) ? 1 : 0;
            dst[i] = (VT)x;
        }
        tbad[(size_t)t] = bd; tirr[(size_t)t] = ir;
    });
    *bad = 0; *regular = true;
    for (int x : tbad) *bad |= x;
    for (int x : tirr) if (x) *regular = false;
}

extern "C" {


int nsk_state_upload(nsk_graph *g, const int64_t *var_value, const int64_t *var_value_evid,
                     const double *weight_value, const int64_t *count) {
    if (!g) return fail(NSK_E_INVALID, "null graph");
    HIPCHECK(hipSetDevice(g->device));
    { int frc = nsk_p2p_flush(g); if (frc) return frc; }
    const int64_t nvar = g->c.nvar;
    const size_t vb = (size_t)g->c.vbytes;
    const int64_t *srcs[2] = {var_value, var_value_evid};
    void *dsts[2] = {g->val, g->val_evid};
    if ((var_value || var_value_evid) && nvar) {
        int rc = xfer_ensure(g);
        if (rc) return rc;
        // validate BOTH chains before anything is copied: an error must leave the device state as it was
        bool regular[2] = {g->chain_regular[0], g->chain_regular[1]};
        for (int k = 0; k < 2; k++) {
            if (!srcs[k]) continue;
            int bad = 0;
            bool reg = true;
            if (vb == 1) narrow_values<int8_t>(g, srcs[k], (int8_t *)g->xfer_host[k], &bad, &reg);
            else narrow_values<int32_t>(g, srcs[k], (int32_t *)g->xfer_host[k], &bad, &reg);
            if (bad) return fail(NSK_E_RANGE, "variable value does not fit the device value type");
            // UFO (inference.py:398-405) uses the first member's value as an index into the factor's
            // member list: the reference reads a neighbouring factor's edge (or faults); refuse
            if (!reg && g->c.has_ufo)
                return fail(NSK_E_RANGE, "a variable value lies outside its domain on a graph with UFO factors "
                                         "(the value indexes the factor's member list)");
            regular[k] = reg;
        }
        const int nb = (int)std::min<int64_t>(4096, (nvar + NSK_BLOCK - 1) / NSK_BLOCK);
        for (int k = 0; k < 2; k++) {
            if (!srcs[k]) continue;
            g->chain_regular[k] = regular[k];
            g->values_regular = g->chain_regular[0] && g->chain_regular[1];
            HIPCHECK(hipMemcpyAsync(g->xfer_dev, g->xfer_host[k], (size_t)nvar * vb, hipMemcpyHostToDevice, g->stream));
            if (vb == 1) k_state_scatter<int8_t><<<dim3(nb), dim3(NSK_BLOCK), 0, g->stream>>>((int8_t *)dsts[k], g->xfer_iid, (const int8_t *)g->xfer_dev, nvar);
            else k_state_scatter<int32_t><<<dim3(nb), dim3(NSK_BLOCK), 0, g->stream>>>((int32_t *)dsts[k], g->xfer_iid, (const int32_t *)g->xfer_dev, nvar);
        }
    }
    if (weight_value && g->c.nweight) {
        const double *src = weight_value;
        if (!g->c.wuser.empty()) {      // the device table is in slot order (nsk_compile.h wmap)
            g->w_stage.resize((size_t)g->c.nweight);
            const int32_t *wuser = g->c.wuser.data();
            double *st = g->w_stage.data();
            nsk::parallel_for(g->c.nweight, [&](int64_t b0, int64_t b1, int) { for (int64_t i = b0; i < b1; i++) st[i] = weight_value[wuser[i]]; });
            src = st;
        }
        HIPCHECK(hipMemcpyAsync(g->w, src, (size_t)g->c.nweight * sizeof(double), hipMemcpyHostToDevice, g->stream));
        if (src != weight_value) HIPCHECK(hipStreamSynchronize(g->stream));
        g->weights_dirty = true;
    }
    if (count && g->c.ncount) {
        HIPCHECK(hipMemcpyAsync(g->cnt_total, count, (size_t)g->c.ncount * sizeof(int64_t), hipMemcpyHostToDevice, g->stream));
        HIPCHECK(hipMemsetAsync(g->cnt, 0, (size_t)g->c.ncount * sizeof(int32_t), g->stream));
        if (g->c.npos) HIPCHECK(hipMemsetAsync(g->cnt_pos, 0, (size_t)g->c.npos, g->stream));
        g->pos_tally_sweeps = 0;
        g->cnt_dirty = false;
    }
    HIPCHECK(hipStreamSynchronize(g->stream));
    return NSK_OK;
}

// device values of one chain -> the caller's int64 array: gather into the caller's order on the device, one
// narrow copy over PCIe, widened by the host threads
static int download_values(nsk_graph *g, const void *src, int64_t *dst, int k) {
    const int64_t nvar = g->c.nvar;
    if (!nvar) return NSK_OK;
    int rc = xfer_ensure(g);
    if (rc) return rc;
    const size_t vb = (size_t)g->c.vbytes;
    const int nb = (int)std::min<int64_t>(4096, (nvar + NSK_BLOCK - 1) / NSK_BLOCK);
    if (vb == 1) k_state_gather<int8_t><<<dim3(nb), dim3(NSK_BLOCK), 0, g->stream>>>((const int8_t *)src, g->xfer_iid, (int8_t *)g->xfer_dev, nvar);
    else k_state_gather<int32_t><<<dim3(nb), dim3(NSK_BLOCK), 0, g->stream>>>((const int32_t *)src, g->xfer_iid, (int32_t *)g->xfer_dev, nvar);
    HIPCHECK(hipMemcpyAsync(g->xfer_host[k], g->xfer_dev, (size_t)nvar * vb, hipMemcpyDeviceToHost, g->stream));
    HIPCHECK(hipStreamSynchronize(g->stream));
    const void *h = g->xfer_host[k];
    nsk::parallel_for(nvar, [&](int64_t b0, int64_t b1, int) {
        if (vb == 1) for (int64_t i = b0; i < b1; i++) dst[i] = ((const int8_t *)h)[i];
        else for (int64_t i = b0; i < b1; i++) dst[i] = ((const int32_t *)h)[i];
    });
    return NSK_OK;
}

int nsk_state_download(nsk_graph *g, int64_t *var_value, int64_t *var_value_evid, double *weight_value,
                       int64_t *count) {
    if (!g) return fail(NSK_E_INVALID, "null graph");
    HIPCHECK(hipSetDevice(g->device));
    { int frc = nsk_p2p_flush(g); if (frc) return frc; }
    int rc;
    if (var_value && (rc = download_values(g, g->val, var_value, 0))) return rc;
    if (var_value_evid && (rc = download_values(g, g->val_evid, var_value_evid, 1))) return rc;
    if (weight_value && g->c.nweight) {
        if (!g->c.wuser.empty()) {      // slot order on the device, the caller's order in `weight_value`
            g->w_stage.resize((size_t)g->c.nweight);
            HIPCHECK(hipMemcpyAsync(g->w_stage.data(), g->w, (size_t)g->c.nweight * sizeof(double), hipMemcpyDeviceToHost, g->stream));
            HIPCHECK(hipStreamSynchronize(g->stream));
            const int32_t *wmap = g->c.wmap.data();
            const double *st = g->w_stage.data();
            nsk::parallel_for(g->c.nweight, [&](int64_t b0, int64_t b1, int) { for (int64_t i = b0; i < b1; i++) weight_value[i] = st[wmap[i]]; });
        } else {
            HIPCHECK(hipMemcpyAsync(weight_value, g->w, (size_t)g->c.nweight * sizeof(double), hipMemcpyDeviceToHost, g->stream));
        }
    }
    if (count) {
        if ((rc = fold_counts(g))) return rc;
        const int64_t nc = g->c.ncount;
        bool done = false;
        if (nc >= (1 << 16)) {          // large tallies cross PCIe as int32 (pinned staging, widened by the host threads)
            if (!g->cnt_dev32) {
                HIPCHECK(hipHostMalloc(&g->cnt_host, (size_t)nc * 4, hipHostMallocNonCoherent));
                if ((rc = dev_alloc(g, &g->cnt_dev32, (size_t)nc))) return rc;
                if ((rc = dev_alloc(g, &g->cnt_wide, 1))) return rc;
            }
            unsigned int wide = 0;
            HIPCHECK(hipMemsetAsync(g->cnt_wide, 0, sizeof(unsigned int), g->stream));
            k_count_narrow<<<dim3((unsigned)std::min<int64_t>(4096, (nc + NSK_BLOCK - 1) / NSK_BLOCK)), dim3(NSK_BLOCK), 0, g->stream>>>(
                g->cnt_total, g->cnt_dev32, nc, g->cnt_wide);
            HIPCHECK(hipMemcpyAsync(g->cnt_host, g->cnt_dev32, (size_t)nc * 4, hipMemcpyDeviceToHost, g->stream));
            HIPCHECK(hipMemcpyAsync(&wide, g->cnt_wide, sizeof(wide), hipMemcpyDeviceToHost, g->stream));
            HIPCHECK(hipStreamSynchronize(g->stream));
            if (!wide) {
                const int32_t *h32 = (const int32_t *)g->cnt_host;
                nsk::parallel_for(nc, [&](int64_t b0, int64_t b1, int) { for (int64_t i = b0; i < b1; i++) count[i] = h32[i]; });
                done = true;
            }
        }
        if (!done && nc)
            HIPCHECK(hipMemcpyAsync(count, g->cnt_total, (size_t)nc * sizeof(int64_t), hipMemcpyDeviceToHost, g->stream));
    }
    // a peer-to-peer exchange that timed out since the last check left ghost values (and merged weights)
    // incomplete: the state handed back is then not a result -- say so here too, not only in nsk_p2p_check
    unsigned int p2p_err = 0;
    if (g->p2p_err) HIPCHECK(hipMemcpyAsync(&p2p_err, g->p2p_err, sizeof(p2p_err), hipMemcpyDeviceToHost, g->stream));
    HIPCHECK(hipStreamSynchronize(g->stream));
    if (p2p_err) {
        HIPCHECK(hipMemsetAsync(g->p2p_err, 0, sizeof(unsigned int), g->stream));          // reported once
        return fail(NSK_E_DEVICE, "peer-to-peer exchange: a peer's boundary values did not arrive within "
                                  "NSK_P2P_TIMEOUT_S; the downloaded state is incomplete");
    }
    return NSK_OK;
}

// 64-bit hash of the compiled layout (nsk_graph_info.layout_hash): FNV-1a over 8-byte words, one lane per array,
// the lanes folded in a fixed order
#define hash_array(v) hash_bytes((v).data(), (v).size() * sizeof((v)[0]))
static uint64_t hash_bytes(const void *data, size_t n) {
    const unsigned char *p = (const unsigned char *)data;
    uint64_t h = 0xcbf29ce484222325ull ^ (uint64_t)n;
    size_t i = 0;
    for (; i + 8 <= n; i += 8) { uint64_t w; memcpy(&w, p + i, 8); h = (h ^ w) * 0x100000001b3ull; h ^= h >> 29; }
    for (; i < n; i++) h = (h ^ p[i]) * 0x100000001b3ull;
    return h;
}
static int64_t layout_hash(const Compiled &c) {
    const uint64_t parts[] = {
        hash_array(c.color), hash_array(c.phase_start), hash_array(c.phase_end), hash_array(c.phase_fast_end),
        hash_array(c.phase_wb_base), hash_array(c.tiles), hash_array(c.tile_wrow), hash_array(c.adj),
        hash_array(c.hub_desc), hash_array(c.hub_adj), hash_array(c.phase_hub_base), hash_array(c.bighub_pos),
        hash_array(c.tile_hdr), hash_array(c.dyn_tiles), hash_array(c.seg_aff), hash_array(c.rest_tiles),
        hash_array(c.learn_rest_tiles), hash_array(c.phase_gen_tile), hash_array(c.ep_desc), hash_array(c.ep_adj),
        hash_array(c.ep_wrow), hash_array(c.ep_win), hash_array(c.ep_win_off), hash_array(c.ep_kstat), hash_array(c.phase_ep_base), hash_array(c.phase_ep),
        hash_array(c.p_vid), hash_array(c.p_slot), hash_array(c.p_cnt), hash_array(c.p_info), hash_array(c.p_init),
        hash_array(c.slot_off), hash_array(c.fidx), hash_array(c.gstream), hash_array(c.gs_off), hash_array(c.f_rec),
        hash_array(c.m_rec), hash_array(c.iid), hash_array(c.w_init), hash_array(c.w_fixed), hash_array(c.w_direct),
        hash_array(c.multi_wids), hash_array(c.wmap), hash_array(c.ghost_needs), hash_array(c.seg_wide), hash_array(c.wide_exc)};
    uint64_t h = 0x9e3779b97f4a7c15ull;
    for (uint64_t x : parts) { h = (h ^ x) * 0x100000001b3ull; h ^= h >> 31; }
    return (int64_t)(h >> 1);             // non-negative
}

static void fill_info(const Compiled &c, nsk_graph_info *info) {
    info->nvar = c.nvar;
    info->nowned = c.nsampled;
    info->ncolors = (int64_t)c.phase_start.size() - 1;
    info->value_bytes = c.vbytes;
    info->device_bytes = 0;
    info->nfast = c.nfast;
    info->ngeneric = c.nsampled - c.nfast;
    info->alg_bytes_inference = c.alg_bytes_inference;
    info->alg_bytes_learning = c.alg_bytes_learning;
    info->sweeps_done = 0;
    info->layout_bytes_inference = c.layout_bytes_inference;
    info->layout_bytes_learning = c.layout_bytes_learning;
    info->ztab_entries = c.nztab;
    info->compile_seconds = 0;
    info->learn_cap = 0.5;
    info->learn_clipped = 0;
    info->grad_shift = c.grad_shift;
    info->acc_copies = 0;
    info->learn_lag = (c.nweight > 0 && c.nweight <= NSK_SMALLW) ? 1 : 0;
    info->direct_weights = c.ndirect;
    info->weight_slots = c.wmap.empty() ? 0 : 1;
    info->layout_hash = getenv("NSK_LAYOUT_HASH") ? layout_hash(c) : 0;
    info->p2p_fused = 0;
    info->tab_quads = c.ntab_quads;
    info->wide_quads = c.nwide_quads;
}

int nsk_graph_get_info(nsk_graph *g, nsk_graph_info *info) {
    if (!g || !info) return fail(NSK_E_INVALID, "null argument");
    fill_info(g->c, info);
    info->device_bytes = g->device_bytes;
    info->sweeps_done = g->sweeps_done;
    info->learn_cap = g->learn_cap;
    if (g->clip_count) {
        unsigned int n = 0;
        (void)hipSetDevice(g->device);
        (void)hipStreamSynchronize(g->stream);
        if (hipMemcpy(&n, g->clip_count, sizeof(n), hipMemcpyDeviceToHost) == hipSuccess) info->learn_clipped = (int64_t)n;
    }
    info->compile_seconds = g->compile_seconds;
    info->acc_copies = g->acc_copies + (g->bins_xcd ? 16 : 0);
    info->learn_lag = (g->learn_lag && g->smallw) ? 1 : 0;
    info->p2p_fused = g->p2p_fused ? 1 : 0;
    return NSK_OK;
}

int nsk_graph_plan(const nsk_graph_desc *desc, int32_t *color, nsk_graph_info *info) {
    if (!desc) return fail(NSK_E_INVALID, "null argument");
    Compiled c;
    std::string err;
    int rc = compile_graph(desc, c, err);
    if (rc) return fail(rc, err);
    if (color && c.nvar) memcpy(color, c.color.data(), (size_t)c.nvar * sizeof(int32_t));
    if (info) fill_info(c, info);
    return NSK_OK;
}

int nsk_graph_plan_needs(const nsk_graph_desc *desc, int64_t *count, int32_t *vids) {
    if (!desc || !count) return fail(NSK_E_INVALID, "null argument");
    Compiled c;
    std::string err;
    int rc = compile_graph(desc, c, err);
    if (rc) return fail(rc, err);
    *count = (int64_t)c.ghost_needs.size();
    if (vids && *count) memcpy(vids, c.ghost_needs.data(), (size_t)*count * sizeof(int32_t));
    return NSK_OK;
}

int nsk_graph_get_layout(nsk_graph *g, int32_t *iid, int64_t *nid) {
    if (!g) return fail(NSK_E_INVALID, "null argument");
    if (iid && g->c.nvar) memcpy(iid, g->c.iid.data(), (size_t)g->c.nvar * sizeof(int32_t));
    if (nid) *nid = g->c.nid;
    return NSK_OK;
}

int nsk_graph_get_generators(nsk_graph *g, int64_t *gen) {
    if (!g || !gen) return fail(NSK_E_INVALID, "null argument");
    const Compiled &c = g->c;
    std::vector<uint8_t> quad((size_t)c.npos, 0);
    for (const Compiled::Segment &sg : c.segments)
        if (sg.ztab >= 0) {
            for (int64_t p = sg.pos0; p < sg.pos0 + (int64_t)sg.ntiles * 64 && p < c.npos; p++) quad[(size_t)p] = 1;
            if (sg.wide < 0) continue;
            const int stride = NSK_WIDE_STRIDE(sg.nslots > 4 ? 2 : 1);
            const int64_t q0 = sg.pos0 >> 8, nq = ((sg.pos0 + 64 * (int64_t)sg.ntiles + 255) >> 8) - q0;
            for (int64_t qi = 0; qi < nq; qi++)
                if (c.seg_wide[(size_t)sg.wide + (size_t)qi * stride] != 0xFFFFFFFFu)
                    for (int64_t p = (q0 + qi) << 8; p < ((q0 + qi + 1) << 8) && p < c.npos; p++) quad[(size_t)p] = 2;
        }
    for (int64_t v = 0; v < c.nvar; v++) {
        const int64_t p = c.color[v] >= 0 ? (int64_t)c.iid[v] : -1;
        const int sch = (p >= 0 && p < c.npos) ? quad[(size_t)p] : 0;
        gen[v] = p < 0 ? -1 : (p | (sch == 1 ? (1ll << 40) : sch == 2 ? (1ll << 41) : 0ll));
    }
    return NSK_OK;
}

int nsk_graph_get_weight_slots(nsk_graph *g, int64_t *slot) {
    if (!g || (!slot && g->c.nweight)) return fail(NSK_E_INVALID, "null argument");
    const Compiled &c = g->c;
    for (int64_t w = 0; w < c.nweight; w++) slot[w] = c.wmap.empty() ? w : (int64_t)c.wmap[(size_t)w];
    return c.wmap.empty() ? 0 : 1;
}

int nsk_graph_get_colors(nsk_graph *g, int32_t *color) {
    if (!g || !color) return fail(NSK_E_INVALID, "null argument");
    if (g->c.nvar) memcpy(color, g->c.color.data(), (size_t)g->c.nvar * sizeof(int32_t));
    return NSK_OK;
}

int nsk_profile_begin(nsk_graph *g) {
    if (!g) return fail(NSK_E_INVALID, "null graph");
    HIPCHECK(hipSetDevice(g->device));
    g->launches_at_begin = g->launches;
    HIPCHECK(hipEventRecord(g->ev0, g->stream));
    return NSK_OK;
}

int nsk_profile_end(nsk_graph *g, double *elapsed_ms, int64_t *kernel_launches) {
    if (!g) return fail(NSK_E_INVALID, "null graph");
    HIPCHECK(hipSetDevice(g->device));
    HIPCHECK(hipEventRecord(g->ev1, g->stream));
    HIPCHECK(hipEventSynchronize(g->ev1));
    float ms = 0.f;
    HIPCHECK(hipEventElapsedTime(&ms, g->ev0, g->ev1));
    if (elapsed_ms) *elapsed_ms = (double)ms;
    if (kernel_launches) *kernel_launches = g->launches - g->launches_at_begin;
    return NSK_OK;
}

int nsk_profile_mark(nsk_graph *g) {
    if (!g) return fail(NSK_E_INVALID, "null graph");
    HIPCHECK(hipSetDevice(g->device));
    HIPCHECK(hipEventRecord(g->ev1, g->stream));
    return NSK_OK;
}

int nsk_profile_read(nsk_graph *g, double *elapsed_ms, int64_t *kernel_launches) {
    if (!g) return fail(NSK_E_INVALID, "null graph");
    HIPCHECK(hipSetDevice(g->device));
    HIPCHECK(hipEventSynchronize(g->ev1));
    float ms = 0.f;
    HIPCHECK(hipEventElapsedTime(&ms, g->ev0, g->ev1));
    if (elapsed_ms) *elapsed_ms = (double)ms;
    if (kernel_launches) *kernel_launches = g->launches - g->launches_at_begin;
    return NSK_OK;
}

int nsk_device_buffer(nsk_graph *g, int which, void **ptr, int64_t *nbytes) {
    if (!g || !ptr) return fail(NSK_E_INVALID, "null argument");
    switch (which) {
    case NSK_BUF_VALUE: *ptr = g->val; if (nbytes) *nbytes = g->c.nid * g->c.vbytes; return NSK_OK;
    case NSK_BUF_VALUE_EVID: *ptr = g->val_evid; if (nbytes) *nbytes = g->c.nid * g->c.vbytes; return NSK_OK;
    case NSK_BUF_WEIGHT: *ptr = g->w; if (nbytes) *nbytes = g->c.nweight * 8; g->weights_exposed = true; return NSK_OK;
    case NSK_BUF_SEND: *ptr = g->x_send; if (nbytes) *nbytes = g->xslot * g->c.vbytes; return NSK_OK;
    case NSK_BUF_RECV: *ptr = g->x_recv; if (nbytes) *nbytes = g->xslot * g->c.vbytes * g->xworld; return NSK_OK;
    case NSK_BUF_SEND_EVID: *ptr = g->x_send_evid; if (nbytes) *nbytes = g->xslot * g->c.vbytes; return NSK_OK;
    case NSK_BUF_RECV_EVID: *ptr = g->x_recv_evid; if (nbytes) *nbytes = g->xslot * g->c.vbytes * g->xworld; return NSK_OK;
    default: return fail(NSK_E_INVALID, "unknown buffer id");
    }
}

int nsk_selftest_exp(int device, const double *x, double *y, int64_t n) {
    if (n < 0 || (n && (!x || !y))) return fail(NSK_E_INVALID, "bad argument");
    if (n == 0) return NSK_OK;
    HIPCHECK(hipSetDevice(device));
    double *dx = nullptr, *dy = nullptr;
    HIPCHECK(hipMalloc((void **)&dx, n * sizeof(double)));
    HIPCHECK(hipMalloc((void **)&dy, n * sizeof(double)));
    HIPCHECK(hipMemcpy(dx, x, n * sizeof(double), hipMemcpyHostToDevice));
    k_selftest_exp<<<dim3((unsigned)((n + 255) / 256)), dim3(256)>>>(dx, dy, n);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipMemcpy(y, dy, n * sizeof(double), hipMemcpyDeviceToHost));
    (void)hipFree(dx); (void)hipFree(dy);
    return NSK_OK;
}

int nsk_selftest_philox(int device, uint64_t seed, uint64_t sweep, uint32_t stream, int64_t n,
                        uint32_t *out) {
    if (n < 0 || (n && !out)) return fail(NSK_E_INVALID, "bad argument");
    if (n == 0) return NSK_OK;
    HIPCHECK(hipSetDevice(device));
    uint32_t *d = nullptr;
    HIPCHECK(hipMalloc((void **)&d, 4 * n * sizeof(uint32_t)));
    k_selftest_philox<<<dim3((unsigned)((n + 255) / 256)), dim3(256)>>>(
        (uint32_t)seed, (uint32_t)(seed >> 32), stream, (uint32_t)sweep, (uint32_t)(sweep >> 32), n, d);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipMemcpy(out, d, 4 * n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    (void)hipFree(d);
    return NSK_OK;
}

int nsk_selftest_stream(int device, int64_t nbytes, int width, int iters, double *gbytes_per_s) {
    // width 4 / 16: plain grid-stride copy (the counter-calibration workload); width 64: the
    // bandwidth ceiling -- 4 x 16-byte non-temporal loads in flight per lane, non-temporal stores
    if (nbytes < 4096 || (width != 4 && width != 16 && width != 64) || iters < 1 || !gbytes_per_s) return fail(NSK_E_INVALID, "bad argument");
    HIPCHECK(hipSetDevice(device));
    nbytes &= ~(int64_t)4095;
    void *a = nullptr, *b = nullptr;
    HIPCHECK(hipMalloc(&a, nbytes));
    HIPCHECK(hipMalloc(&b, nbytes));
    HIPCHECK(hipMemset(a, 1, nbytes));
    hipEvent_t e0, e1;
    HIPCHECK(hipEventCreate(&e0));
    HIPCHECK(hipEventCreate(&e1));
    const long long n = nbytes / (width == 64 ? 16 : width);
    const int grid = 256 * 8;
    for (int it = -1; it < iters; it++) {
        if (it == 0) HIPCHECK(hipEventRecord(e0, 0));
        if (width == 4) k_stream_copy<uint32_t><<<dim3(grid), dim3(NSK_BLOCK)>>>((const uint32_t *)a, (uint32_t *)b, n);
        else if (width == 16) k_stream_copy<uint4><<<dim3(grid), dim3(NSK_BLOCK)>>>((const uint4 *)a, (uint4 *)b, n);
        else k_stream_copy_nt<4><<<dim3(grid * 2), dim3(NSK_BLOCK)>>>((const uint4 *)a, (uint4 *)b, n);
    }
    HIPCHECK(hipEventRecord(e1, 0));
    HIPCHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
    *gbytes_per_s = 2.0 * (double)nbytes * iters / ((double)ms * 1e-3) / 1e9;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(a); (void)hipFree(b);
    return NSK_OK;
}

int nsk_ghost_needs(nsk_graph *g, int64_t *count, int32_t *vids) {
    if (!g || !count) return fail(NSK_E_INVALID, "null argument");
    *count = (int64_t)g->c.ghost_needs.size();
    if (vids && *count) memcpy(vids, g->c.ghost_needs.data(), (size_t)*count * sizeof(int32_t));
    return NSK_OK;
}

int nsk_exchange_setup(nsk_graph *g, int world, int rank, const int32_t *send_vids, int64_t nsend,
                       const int32_t *recv_vids, const int64_t *recv_off, int64_t slot) {
    if (!g || world < 1 || rank < 0 || rank >= world || nsend < 0 || slot < nsend || !recv_off)
        return fail(NSK_E_INVALID, "bad exchange description");
    HIPCHECK(hipSetDevice(g->device));
    const int64_t nrecv = recv_off[world];
    for (int64_t i = 0; i < nsend; i++)
        if (send_vids[i] < g->c.own_begin || send_vids[i] >= g->c.own_end)
            return fail(NSK_E_INDEX, "send list names a variable this handle does not own");
    std::vector<int32_t> rslot((size_t)nrecv);
    for (int src = 0; src < world; src++) {
        if (recv_off[src + 1] - recv_off[src] > slot) return fail(NSK_E_INVALID, "slot smaller than a rank's list");
        for (int64_t j = recv_off[src]; j < recv_off[src + 1]; j++) {
            if (recv_vids[j] < -1 || recv_vids[j] >= g->c.nvar) return fail(NSK_E_INDEX, "receive list out of range");
            rslot[j] = (src == rank || recv_vids[j] < 0) ? -1 : (int32_t)(src * slot + (j - recv_off[src]));
        }
    }
    g->xworld = world; g->xrank = rank; g->xslot = slot; g->xnsend = nsend; g->xnrecv = nrecv;
    std::vector<int32_t> sv(send_vids, send_vids + nsend), rv(recv_vids, recv_vids + nrecv);
    for (auto &x : sv) x = g->c.iid[x];                        // the kernels address values by internal id
    for (auto &x : rv) x = x < 0 ? 0 : g->c.iid[x];          // (skipped entries: slot -1, never written)
    int rc;
    if ((rc = dev_upload(g, &g->x_send_vids, sv))) return rc;
    if ((rc = dev_upload(g, &g->x_recv_vids, rv))) return rc;
    if ((rc = dev_upload(g, &g->x_recv_slot, rslot))) return rc;
    const size_t vb = (size_t)g->c.vbytes;
    uint8_t *t = nullptr;
    if ((rc = dev_alloc(g, &t, (size_t)slot * vb))) return rc; g->x_send = t;
    if ((rc = dev_alloc(g, &t, (size_t)slot * vb * world))) return rc; g->x_recv = t;
    if ((rc = dev_alloc(g, &t, (size_t)slot * vb))) return rc; g->x_send_evid = t;
    if ((rc = dev_alloc(g, &t, (size_t)slot * vb * world))) return rc; g->x_recv_evid = t;
    HIPCHECK(hipMemsetAsync(g->x_send, 0, (size_t)(slot ? slot : 1) * vb, g->stream));
    HIPCHECK(hipMemsetAsync(g->x_send_evid, 0, (size_t)(slot ? slot : 1) * vb, g->stream));
    if (!g->w_start) {
        if ((rc = dev_alloc(g, &g->w_start, (size_t)g->c.nweight))) return rc;
        if ((rc = dev_alloc(g, &g->w_delta, (size_t)g->c.nweight))) return rc;
    }
    HIPCHECK(hipStreamSynchronize(g->stream));
    return NSK_OK;
}

}  // extern "C"

void nsk_drop_sweep_graph(nsk_graph *g) {
    if (g->sweep_graph) { (void)hipGraphExecDestroy(g->sweep_graph); g->sweep_graph = nullptr; }
    g->sweep_graph_key = -1;
    if (g->sweep_graph_big) { (void)hipGraphExecDestroy(g->sweep_graph_big); g->sweep_graph_big = nullptr; }
    g->sweep_graph_big_key = -1;
}

template <typename VT>
static int exchange_kernels(nsk_graph *g, int which, bool pack) {
    VT *val = (VT *)(which == NSK_BUF_VALUE ? g->val : g->val_evid);
    VT *sb = (VT *)(which == NSK_BUF_VALUE ? g->x_send : g->x_send_evid);
    VT *rb = (VT *)(which == NSK_BUF_VALUE ? g->x_recv : g->x_recv_evid);
    if (pack) {
        const int n = (int)g->xnsend;
        if (n > 0)
            k_exchange_pack<VT><<<dim3((n + NSK_BLOCK - 1) / NSK_BLOCK), dim3(NSK_BLOCK), 0, g->stream>>>(
                val, g->x_send_vids, sb, n);
    } else {
        const int n = (int)g->xnrecv;
        if (n > 0)
            k_exchange_unpack<VT><<<dim3((n + NSK_BLOCK - 1) / NSK_BLOCK), dim3(NSK_BLOCK), 0, g->stream>>>(
                val, g->x_recv_vids, g->x_recv_slot, rb, n);
    }
    HIPCHECK(hipGetLastError());
    return NSK_OK;
}

extern "C" {

static int exchange_step(nsk_graph *g, int which, bool pack) {
    if (!g) return fail(NSK_E_INVALID, "null graph");
    if (g->xworld == 0) return fail(NSK_E_INVALID, "nsk_exchange_setup has not been called");
    if (which != NSK_BUF_VALUE && which != NSK_BUF_VALUE_EVID) return fail(NSK_E_INVALID, "bad buffer id");
    HIPCHECK(hipSetDevice(g->device));
    { int frc = nsk_p2p_flush(g); if (frc) return frc; }
    return g->c.vbytes == 1 ? exchange_kernels<int8_t>(g, which, pack) : exchange_kernels<int32_t>(g, which, pack);
}

int nsk_exchange_pack(nsk_graph *g, int which) { return exchange_step(g, which, true); }
int nsk_exchange_unpack(nsk_graph *g, int which) { return exchange_step(g, which, false); }

// ---- peer-to-peer exchange ------------------------------------------------------------------------
// Partial factors (messages.py:1333-1355): `npf` aggregates over variables this handle holds; op 0 = "some member is
// 1" (OR), 1 = "no member is 0" (AND / ISTRUE); members of aggregate j = member_vids[member_off[j] .. member_off[j+1]).
// The value arrays grow by npf slots behind the internal ids; a peer-to-peer send list names aggregate j as variable
// id nvar + j, and every exchange recomputes the aggregates (both chains in learning) before it pushes.
int nsk_pf_setup(nsk_graph *g, int64_t npf, const uint8_t *op, const int64_t *member_off, const int32_t *member_vids) {
    if (!g || npf < 0 || (npf && (!op || !member_off || !member_vids))) return fail(NSK_E_INVALID, "bad partial-factor description");
    HIPCHECK(hipSetDevice(g->device));
    { int frc = nsk_p2p_flush(g); if (frc) return frc; }
    std::vector<int32_t> off((size_t)npf + 1, 0), mem;
    std::vector<uint8_t> ops((size_t)npf);
    for (int64_t j = 0; j < npf; j++) {
        if (op[j] > 1 || member_off[j + 1] < member_off[j] || member_off[0] != 0) return fail(NSK_E_INVALID, "bad partial-factor description");
        ops[(size_t)j] = op[j];
        for (int64_t k = member_off[j]; k < member_off[j + 1]; k++) {
            if (member_vids[k] < 0 || member_vids[k] >= g->c.nvar) return fail(NSK_E_INDEX, "partial factor over a variable this handle does not hold");
            mem.push_back(g->c.iid[member_vids[k]]);
        }
        off[(size_t)j + 1] = (int32_t)mem.size();
    }
    HIPCHECK(hipStreamSynchronize(g->stream));
    nsk_drop_sweep_graph(g);
    // value arrays with npf more slots (contents kept)
    const size_t vb = (size_t)g->c.vbytes, nid = (size_t)g->c.nid;
    for (int chain = 0; chain < 2; chain++) {
        void *&arr = chain ? g->val_evid : g->val;
        uint8_t *bigger = nullptr;
        int rc = dev_alloc(g, &bigger, (nid + (size_t)npf) * vb + 16);
        if (rc) return rc;
        HIPCHECK(hipMemsetAsync(bigger, 0, (nid + (size_t)npf) * vb + 16, g->stream));
        HIPCHECK(hipMemcpyAsync(bigger, arr, nid * vb, hipMemcpyDeviceToDevice, g->stream));
        HIPCHECK(hipStreamSynchronize(g->stream));
        dev_free(g, arr);
        arr = bigger;
    }
    // (a second set-up replaces the first one's descriptions)
    dev_free(g, g->pf_op); g->pf_op = nullptr;
    dev_free(g, g->pf_off); g->pf_off = nullptr;
    dev_free(g, g->pf_mem); g->pf_mem = nullptr;
    int rc;
    if ((rc = dev_upload(g, &g->pf_op, ops))) return rc;
    if ((rc = dev_upload(g, &g->pf_off, off))) return rc;
    if ((rc = dev_upload(g, &g->pf_mem, mem))) return rc;
    HIPCHECK(hipStreamSynchronize(g->stream));
    g->npf = npf;
    return NSK_OK;
}

int nsk_p2p_setup(nsk_graph *g, int world, int rank, const int32_t *send_vids, const int64_t *send_off,
                  const int32_t *recv_vids, const int64_t *recv_off, const int64_t *peer_base,
                  const int64_t *peer_total) {
    if (!g || world < 1 || world > 16 || rank < 0 || rank >= world || !send_off || !recv_off || !peer_base || !peer_total)
        return fail(NSK_E_INVALID, "bad peer-to-peer description (at most 16 ranks: one node)");
    HIPCHECK(hipSetDevice(g->device));
    const int64_t nsend = send_off[world], nrecv = recv_off[world];
    if (send_off[0] != 0 || recv_off[0] != 0 || nsend < 0 || nrecv < 0) return fail(NSK_E_INVALID, "bad list offsets");
    for (int q = 0; q < world; q++) {
        if (send_off[q + 1] < send_off[q] || recv_off[q + 1] < recv_off[q]) return fail(NSK_E_INVALID, "bad list offsets");
        const int64_t seg = send_off[q + 1] - send_off[q];
        if (peer_base[q] < 0 || peer_base[q] + seg > peer_total[q]) return fail(NSK_E_INVALID, "a send segment does not fit its reader's buffer");
    }
    if (send_off[rank + 1] != send_off[rank] || recv_off[rank + 1] != recv_off[rank])
        return fail(NSK_E_INVALID, "a rank does not exchange with itself");
    if (peer_total[rank] != nrecv) return fail(NSK_E_INVALID, "peer_total[rank] must be this rank's receive total");
    std::vector<int32_t> sv((size_t)nsend), rv((size_t)nrecv);
    for (int64_t i = 0; i < nsend; i++) {
        if (send_vids[i] >= g->c.nvar && send_vids[i] < g->c.nvar + g->npf) {       // partial-factor aggregate (nsk_pf_setup)
            sv[(size_t)i] = (int32_t)(g->c.nid + (send_vids[i] - g->c.nvar));
            continue;
        }
        if (send_vids[i] < g->c.own_begin || send_vids[i] >= g->c.own_end)
            return fail(NSK_E_INDEX, "send list names a variable this handle does not own");
        sv[(size_t)i] = g->c.iid[send_vids[i]];               // the kernels address values by internal id
    }
    for (int64_t j = 0; j < nrecv; j++) {
        if (recv_vids[j] < 0 || recv_vids[j] >= g->c.nvar || (recv_vids[j] >= g->c.own_begin && recv_vids[j] < g->c.own_end))
            return fail(NSK_E_INDEX, "receive list names a variable this handle owns or does not hold");
        rv[(size_t)j] = g->c.iid[recv_vids[j]];
    }
    nsk_drop_sweep_graph(g);
    g->p2p_ready = false;
    g->pworld = world; g->prank = rank; g->p_nsend = nsend; g->p_nrecv = nrecv;
    g->p_soff.assign(send_off, send_off + world + 1);
    g->p_roff.assign(recv_off, recv_off + world + 1);
    g->p_dbase.assign(peer_base, peer_base + world);
    g->p_dtotal.assign(peer_total, peer_total + world);
    g->p2p_peer_mask = 0;
    for (int q = 0; q < world; q++)             // symmetric: q is a peer when either side reads from the other
        if (q != rank && (send_off[q + 1] > send_off[q] || recv_off[q + 1] > recv_off[q])) g->p2p_peer_mask |= 1u << q;
    int rc;
    if ((rc = dev_upload(g, &g->p_send_iid, sv))) return rc;
    if ((rc = dev_upload(g, &g->p_recv_iid, rv))) return rc;
    g->p_send_host = sv;
    g->p_recv_host = rv;
    g->p2p_fused = false;
    if (!g->w_start) {
        if ((rc = dev_alloc(g, &g->w_start, (size_t)g->c.nweight))) return rc;
        if ((rc = dev_alloc(g, &g->w_delta, (size_t)g->c.nweight))) return rc;
    }
    if (!g->p2p_err) {
        if ((rc = dev_alloc(g, &g->p2p_err, 4))) return rc;          // [0] error mark, [1] [2] the kernels' tickets
        HIPCHECK(hipMemsetAsync(g->p2p_err, 0, 4 * sizeof(unsigned int), g->stream));
    }
    if (const char *t = getenv("NSK_P2P_TIMEOUT_S")) {
        const double sec = atof(t);
        if (sec > 0) g->p2p_timeout_ticks = (unsigned long long)(sec * 1e8);
    }
    HIPCHECK(hipStreamSynchronize(g->stream));
    return NSK_OK;
}

static void p2p_close_peers(nsk_graph *g) {
    for (int q = 0; q < 16; q++) {
        if (g->p2p_peer_ipc[q] && g->p2p_peer_base[q]) (void)hipIpcCloseMemHandle(g->p2p_peer_base[q]);
        g->p2p_peer_base[q] = nullptr;
        g->p2p_peer_ipc[q] = false;
    }
}

int nsk_p2p_export(nsk_graph *g, void *handle64, void **base) {
    if (!g) return fail(NSK_E_INVALID, "null argument");
    if (g->pworld == 0) return fail(NSK_E_INVALID, "nsk_p2p_setup has not been called");
    HIPCHECK(hipSetDevice(g->device));
    const size_t vb = (size_t)g->c.vbytes, nw = (size_t)g->c.nweight;
    const size_t bytes = nsk_p2p_bytes(g->pworld, (size_t)g->p_nrecv, vb, nw);
    if (g->p2p_base && g->p2p_bytes < bytes) {          // a later set-up with longer lists: a new allocation
        HIPCHECK(hipStreamSynchronize(g->stream));
        g->allocs.erase(std::remove(g->allocs.begin(), g->allocs.end(), g->p2p_base), g->allocs.end());
        (void)hipFree(g->p2p_base);
        g->device_bytes -= (int64_t)g->p2p_bytes;
        g->p2p_base = nullptr;
    }
    if (!g->p2p_base) {
        // fine-grained: a peer's stores and this rank's polling loads are coherent while kernels run
        HIPCHECK(hipExtMallocWithFlags(&g->p2p_base, bytes, hipDeviceMallocFinegrained));
        g->allocs.push_back(g->p2p_base);
        g->p2p_bytes = bytes;
        g->device_bytes += (int64_t)bytes;
    }
    HIPCHECK(hipMemsetAsync(g->p2p_base, 0, g->p2p_bytes, g->stream));
    HIPCHECK(hipMemsetAsync(g->p2p_err, 0, 4 * sizeof(unsigned int), g->stream));
    HIPCHECK(hipStreamSynchronize(g->stream));
    g->p2p_tag = 0;
    g->p2p_ready = false;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    if (handle64) {
        hipIpcMemHandle_t h;
        HIPCHECK(hipIpcGetMemHandle(&h, g->p2p_base));
        memcpy(handle64, &h, 64);
    }
    if (base) *base = g->p2p_base;
    return NSK_OK;
}

}  // extern "C"

// Does every sampled variable of the handle live in a table segment (the grids and their shards)?
bool nsk_tables_only(const nsk_graph *g) {
    const nsk::Compiled &c = g->c;
    const size_t nphase = c.phase_start.size() - 1;
    if (g->scan != NSK_SCAN_CHROMATIC || nphase == 0) return false;
    for (size_t ph = 0; ph < nphase; ph++) {
        const int64_t ntiles = c.phase_wb_base[ph + 1] - c.phase_wb_base[ph];
        if (c.phase_end[ph] > c.phase_fast_end[ph]) return false;                    // generic-path variables / hubs
        if (ntiles > c.phase_gen_tile[ph]) return false;                             // general tiles
        if (c.phase_rest_base[ph + 1] > c.phase_rest_base[ph]) return false;         // tiles outside segments
    }
    for (const nsk::Compiled::Segment &sg : c.segments) if (sg.ztab < 0) return false;
    return true;
}

// Fused boundary exchange (nsk_internal.h p2p_fused): decide whether the handle qualifies and build the push map.
// Conditions: every sampled variable in a table segment; the receive list is the run of ghost ids in order (the
// compiler numbers the ghosts a handle reads first and ascending, so the receive block IS the ghost array); every
// boundary value has exactly one reader (range shards of a grid; a value read by several ranks keeps the
// exchange kernels).
static int p2p_fuse_plan(nsk_graph *g) {
    g->p2p_fused = false;
    g->p2p_border_tiles.clear();
    if (nsk::diag_env("NSK_NO_P2P_FUSE") || !nsk_tables_only(g)) return NSK_OK;
    const nsk::Compiled &c = g->c;
    const std::vector<int32_t> &sv = g->p_send_host, &rv = g->p_recv_host;
    if (rv.empty() && sv.empty()) return NSK_OK;
    const uint32_t ghost_lo = rv.empty() ? (uint32_t)c.nid : (uint32_t)rv[0];
    for (size_t j = 0; j < rv.size(); j++) if ((uint32_t)rv[j] != ghost_lo + (uint32_t)j) return NSK_OK;
    if (ghost_lo < (uint32_t)c.npos) return NSK_OK;
    for (int q = 0; q < g->pworld; q++) if (g->p_dtotal[q] >= (1ll << 28)) return NSK_OK;
    // tiles that read a ghost: slot bases of the implicit adjacency, or the stream words
    std::vector<int32_t> tiles;
    for (int32_t p : sv) tiles.push_back(p >> 6);
    for (const nsk::Compiled::Segment &sg : c.segments) {
        const int nch = sg.nslots > 4 ? 2 : 1;
        for (int64_t t = 0; t < sg.ntiles; t++) {
            bool reads = false;
            const uint32_t *ab = sg.aff >= 0 ? &c.seg_aff[((size_t)sg.aff + (size_t)t * nch) * 4] : nullptr;
            if (ab && ab[0] != 0xFFFFFFFFu) {
                for (int j = 0; j < 4 * nch && !reads; j++) reads = ab[j] + 63u >= ghost_lo && ab[j] < ghost_lo + (uint32_t)rv.size();
            } else {
                const uint32_t *w = &c.adj[((size_t)sg.adj_off + (size_t)t * 64 * nch) * 4];
                for (int i = 0; i < 256 * nch && !reads; i++) reads = w[i] - ghost_lo < (uint32_t)rv.size();
            }
            if (reads) tiles.push_back((int32_t)(sg.pos0 / 64 + t));
        }
    }
    std::sort(tiles.begin(), tiles.end());
    tiles.erase(std::unique(tiles.begin(), tiles.end()), tiles.end());
    // fewer, longer runs (a launch carries at most NSK_SEG_MAX segment entries): a short segment with a border
    // tile is border as a whole, and so are gaps of a few tiles between border tiles of one segment -- such a tile
    // pushes nothing (its row of the map is empty), it only waits and counts like its neighbours
    {
        std::vector<int32_t> extra;
        for (const nsk::Compiled::Segment &sg : c.segments) {
            const int32_t f = (int32_t)(sg.pos0 / 64), e = f + sg.ntiles;
            auto lo = std::lower_bound(tiles.begin(), tiles.end(), f), hi = std::lower_bound(tiles.begin(), tiles.end(), e);
            if (lo == hi) continue;
            if (sg.ntiles <= 64) { for (int32_t t = f; t < e; t++) extra.push_back(t); continue; }
            for (auto it = lo; it + 1 < hi; ++it)
                if (it[1] - it[0] > 1 && it[1] - it[0] <= 16) for (int32_t t = it[0] + 1; t < it[1]; t++) extra.push_back(t);
        }
        tiles.insert(tiles.end(), extra.begin(), extra.end());
        std::sort(tiles.begin(), tiles.end());
        tiles.erase(std::unique(tiles.begin(), tiles.end()), tiles.end());
    }
    std::vector<uint32_t> pm(tiles.size() * 64, 0xFFFFFFFFu);
    for (int q = 0; q < g->pworld; q++)
        for (int64_t k = g->p_soff[q]; k < g->p_soff[q + 1]; k++) {
            const int32_t pos = sv[(size_t)k];
            const size_t row = (size_t)(std::lower_bound(tiles.begin(), tiles.end(), pos >> 6) - tiles.begin());
            uint32_t &e = pm[row * 64 + (size_t)(pos & 63)];
            if (e != 0xFFFFFFFFu) return NSK_OK;                                     // a second reader
            e = ((uint32_t)q << 28) | (uint32_t)(g->p_dbase[q] + (k - g->p_soff[q]));
        }
    if (g->p2p_push_map) {
        g->allocs.erase(std::remove(g->allocs.begin(), g->allocs.end(), (void *)g->p2p_push_map), g->allocs.end());
        (void)hipFree(g->p2p_push_map);
        g->p2p_push_map = nullptr;
    }
    int rc = dev_upload(g, &g->p2p_push_map, pm);
    if (rc) return rc;
    if (!g->d_counters) {}      // (the border counter lives behind the error mark: p2p_err[3])
    g->p2p_border_tiles.swap(tiles);
    g->p2p_ghost_lo = ghost_lo;
    g->p2p_fused = true;
    g->seg_plans_key = -1;          // the segment plans split at the border tiles
    return NSK_OK;
}

extern "C" {

static int p2p_finish_import(nsk_graph *g) {
    if (g->c.nweight)       // the weights every rank starts the next learning epoch from
        HIPCHECK(hipMemcpyAsync(g->w_start, g->w, (size_t)g->c.nweight * sizeof(double), hipMemcpyDeviceToDevice, g->stream));
    int rc = p2p_fuse_plan(g);
    if (rc) return rc;
    HIPCHECK(hipStreamSynchronize(g->stream));
    nsk_drop_sweep_graph(g);
    g->p2p_tag = 0;
    g->p2p_close_pending = false;
    g->p2p_ready = true;
    return NSK_OK;
}

int nsk_p2p_import(nsk_graph *g, const void *all_handles) {
    if (!g || !all_handles) return fail(NSK_E_INVALID, "null argument");
    if (!g->p2p_base) return fail(NSK_E_INVALID, "nsk_p2p_export first");
    HIPCHECK(hipSetDevice(g->device));
    p2p_close_peers(g);
    for (int q = 0; q < g->pworld; q++) {
        if (q == g->prank) { g->p2p_peer_base[q] = g->p2p_base; continue; }
        hipIpcMemHandle_t h;
        memcpy(&h, (const char *)all_handles + (size_t)q * 64, 64);
        HIPCHECK(hipIpcOpenMemHandle(&g->p2p_peer_base[q], h, hipIpcMemLazyEnablePeerAccess));
        g->p2p_peer_ipc[q] = true;
    }
    return p2p_finish_import(g);
}

int nsk_p2p_import_local(nsk_graph *g, void *const *bases) {
    if (!g || !bases) return fail(NSK_E_INVALID, "null argument");
    if (!g->p2p_base) return fail(NSK_E_INVALID, "nsk_p2p_export first");
    HIPCHECK(hipSetDevice(g->device));
    p2p_close_peers(g);
    for (int q = 0; q < g->pworld; q++) {
        if (q != g->prank && !bases[q]) return fail(NSK_E_INVALID, "null peer allocation");
        g->p2p_peer_base[q] = q == g->prank ? g->p2p_base : bases[q];
    }
    return p2p_finish_import(g);
}

}  // extern "C"

// part 0 = one whole exchange (the sweep loops); 1 = the pushes, 2 = wait + unpack (+ the owner's half of the
// weight merge), 3 = the closing half of the weight merge -- the parts on their own serve the tests that drive
// several handles from one process (issued breadth-first) and the phase timings
template <typename VT>
static int p2p_exchange(nsk_graph *g, const unsigned long long *tag_base, unsigned int tag_off, bool learn, int part, int selftest = 0) {
    const int world = g->pworld, me = g->prank;
    const int nw = (int)g->c.nweight;
    if ((part == 0 || part == 1) && !tag_base) ++g->p2p_tag;
    const unsigned int tag = tag_base ? tag_off : g->p2p_tag;
    const bool weights = learn && nw > 0 && world > 1;
    // a learning epoch's weight deltas go to every rank, so every rank is a peer of every other
    const unsigned int mask = weights ? (((1u << world) - 1u) & ~(1u << me)) : g->p2p_peer_mask;
    if (!mask) return NSK_OK;
    P2PPlan plan;
    memset(&plan, 0, sizeof(plan));
    for (int q = 0; q < world; q++) {
        plan.base[q] = g->p2p_peer_base[q];
        plan.soff[q] = (unsigned long long)g->p_soff[q];
        plan.roff[q] = (unsigned long long)g->p_roff[q];
        plan.dbase[q] = (unsigned long long)g->p_dbase[q];
        plan.dtotal[q] = (unsigned long long)g->p_dtotal[q];
    }
    for (int q = world; q <= 16; q++) { plan.soff[q] = (unsigned long long)g->p_nsend; plan.roff[q] = (unsigned long long)g->p_nrecv; }
    P2PWeights pw;
    // big: lists beyond 2^16 values or tables beyond 2^16 weights -- many-block launches with one-wave flag kernels
    // between them (k_p2p_push_big); otherwise one or two <= 64-block launches that raise and poll themselves
    const char *big_env = nsk::diag_env("NSK_P2P_BIG_MIN");             // (diagnostic; read per exchange so that tests can set it)
    const int64_t big_min = big_env ? atoll(big_env) : 65536;
    const bool big = std::max(g->p_nsend, g->p_nrecv) > big_min || (weights && nw > big_min);
    pw.w = weights ? g->w : nullptr; pw.w_start = weights ? g->w_start : nullptr; pw.nw = weights ? nw : 0;
    const int64_t wwork = weights ? ((int64_t)nw + 3) / 4 : 0;          // (a block's threads take a few weights each)
    auto blocks = [&](int64_t work) { return (int)std::max<int64_t>(1, std::min<int64_t>(64, (work + 4 * NSK_BLOCK - 1) / (4 * NSK_BLOCK))); };
    auto many = [&](int64_t work) { return dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(1024, (work + NSK_BLOCK - 1) / NSK_BLOCK))); };
    VT *val = (VT *)g->val, *val_evid = (VT *)g->val_evid;
    const int both = learn ? 1 : 0;
    auto push_big = [&]() {
        k_p2p_push_big<VT><<<many(std::max<int64_t>(g->p_nsend, pw.nw)), dim3(NSK_BLOCK), 0, g->stream>>>(
            val, val_evid, both, g->p_send_iid, (long long)g->p_nsend, plan, pw, world, me, tag, tag_base, selftest);
        k_p2p_raise<<<dim3(1), dim3(64), 0, g->stream>>>(plan, 0, world, me, mask, tag, tag_base);
    };
    auto unpack_big = [&]() {
        k_p2p_wait<<<dim3(1), dim3(64), 0, g->stream>>>(g->p2p_base, 0, world, mask, tag, tag_base, g->p2p_err, g->p2p_timeout_ticks);
        k_p2p_unpack_big<VT><<<many(std::max<int64_t>(g->p_nrecv, (pw.nw + world - 1) / world)), dim3(NSK_BLOCK), 0, g->stream>>>(
            val, val_evid, both, g->p_recv_iid, (long long)g->p_nrecv, g->p2p_base, plan, pw, world, me, tag, tag_base, g->p2p_err, selftest);
        if (weights) k_p2p_raise<<<dim3(1), dim3(64), 0, g->stream>>>(plan, 1, world, me, mask, tag, tag_base);
    };
    auto gather = [&]() {
        if (!weights) return;
        if (big) k_p2p_wait<<<dim3(1), dim3(64), 0, g->stream>>>(g->p2p_base, 1, world, mask, tag, tag_base, g->p2p_err, g->p2p_timeout_ticks);
        k_p2p_gather_w<VT><<<big ? many(nw) : dim3((unsigned)blocks(wwork)), dim3(NSK_BLOCK), 0, g->stream>>>(
            g->w, g->w_start, nw, g->p2p_base, (long long)g->p_nrecv, world, mask, tag, g->p2p_err, g->p2p_timeout_ticks, selftest, big ? 1 : 0);
        if (!selftest) g->weights_dirty = true;
    };
    if ((part == 0 || part == 1) && g->npf > 0 && !selftest)         // the partial-factor aggregates this rank's readers take
        k_pf_compute<VT><<<dim3((unsigned)((g->npf + NSK_BLOCK - 1) / NSK_BLOCK)), dim3(NSK_BLOCK), 0, g->stream>>>(
            val, val_evid, both, g->pf_op, g->pf_off, g->pf_mem, (int)g->npf, (long long)g->c.nid);
    if (part == 0) {                            // the sweep loops: push, flags, wait and unpack in one launch
        if (big) { push_big(); unpack_big(); }
        else {
            const int nb = blocks(std::max(std::max(g->p_nsend, g->p_nrecv), wwork));
            k_p2p_exchange<VT, true><<<dim3(nb), dim3(NSK_BLOCK), 0, g->stream>>>(
                val, val_evid, both, g->p_send_iid, (long long)g->p_nsend, plan, pw, g->p_recv_iid,
                (long long)g->p_nrecv, g->p2p_base, world, me, mask, g->p2p_err + 1, tag, g->p2p_err, tag_base, g->p2p_timeout_ticks, selftest);
        }
        gather();
    } else if (part == 1) {
        if (big) push_big();
        else {
            // at most 64 blocks (grid-stride): the closing ticket adds must not queue up
            const int nb = blocks(std::max(g->p_nsend, wwork));
            k_p2p_push<VT><<<dim3(nb), dim3(NSK_BLOCK), 0, g->stream>>>(val, val_evid, both, g->p_send_iid, (long long)g->p_nsend, plan, pw, world, me,
                                                                       mask, g->p2p_err + 1, tag, tag_base, selftest);
        }
    } else if (part == 2) {
        if (big) unpack_big();
        else {
            const int nb = blocks(std::max(g->p_nrecv, wwork));
            k_p2p_exchange<VT, false><<<dim3(nb), dim3(NSK_BLOCK), 0, g->stream>>>(
                val, val_evid, both, g->p_send_iid, (long long)g->p_nsend, plan, pw, g->p_recv_iid,
                (long long)g->p_nrecv, g->p2p_base, world, me, mask, g->p2p_err + 1, tag, g->p2p_err, tag_base, g->p2p_timeout_ticks, selftest);
        }
    } else {
        gather();
    }
    HIPCHECK(hipGetLastError());
    return NSK_OK;
}

int nsk_p2p_enqueue(nsk_graph *g, const unsigned long long *tag_base, unsigned int tag_off, bool learn, int part) {
    return g->c.vbytes == 1 ? p2p_exchange<int8_t>(g, tag_base, tag_off, learn, part)
                            : p2p_exchange<int32_t>(g, tag_base, tag_off, learn, part);
}

// kernel argument of a fused table launch
void nsk_p2p_fill(nsk_graph *g, nsk::TabP2P &px, const unsigned long long *tag_base, unsigned int tag, bool wait) {
    memset(&px, 0, sizeof(px));
    px.wait = wait ? 1 : 0;
    px.mine = g->p2p_base;
    for (int q = 0; q < g->pworld; q++) { px.peer[q] = g->p2p_peer_base[q]; px.dtotal[q] = (unsigned long long)g->p_dtotal[q]; }
    px.push_map = g->p2p_push_map;
    px.counter = g->p2p_err + 3;
    px.err = g->p2p_err;
    px.tag_base = tag_base;
    px.timeout_ticks = g->p2p_timeout_ticks;
    px.ghost_lo = g->p2p_ghost_lo;
    px.nrecv = (uint32_t)g->p_nrecv;
    px.border_total = g->p2p_border_total;
    px.tag = tag;
    px.peer_mask = g->p2p_peer_mask;
    px.world = g->pworld;
    px.me = g->prank;
}

// the ghost values of the value array into the receive block of the LAST exchange's parity: what the first fused
// sweep of a call reads (the caller may have uploaded a state since)
template <typename VT>
static __global__ __launch_bounds__(NSK_BLOCK) void k_p2p_ghost_pack(const VT *val, const int32_t *recv_iid, long long nrecv, void *mine,
                                                                      int world, unsigned int tag) {
    VT *rb = (VT *)((char *)mine + nsk_p2p_recv_off(world)) + (size_t)(tag & 1u) * 2 * (size_t)nrecv;
    for (long long j = (long long)blockIdx.x * NSK_BLOCK + threadIdx.x; j < nrecv; j += (long long)gridDim.x * NSK_BLOCK)
        rb[j] = val[recv_iid[j]];
}
int nsk_p2p_ghost_pack(nsk_graph *g) {
    if (g->p_nrecv == 0) return NSK_OK;
    const int nb = (int)std::min<int64_t>(64, (g->p_nrecv + NSK_BLOCK - 1) / NSK_BLOCK);
    if (g->c.vbytes == 1)
        k_p2p_ghost_pack<int8_t><<<dim3(nb), dim3(NSK_BLOCK), 0, g->stream>>>((const int8_t *)g->val, g->p_recv_iid, (long long)g->p_nrecv,
                                                                             g->p2p_base, g->pworld, g->p2p_tag);
    else
        k_p2p_ghost_pack<int32_t><<<dim3(nb), dim3(NSK_BLOCK), 0, g->stream>>>((const int32_t *)g->val, g->p_recv_iid, (long long)g->p_nrecv,
                                                                              g->p2p_base, g->pworld, g->p2p_tag);
    HIPCHECK(hipGetLastError());
    return NSK_OK;
}

// A fused sweep sequence ends with its last class launch; the wait for the peers' last flags and the copy of the
// received values into the value array's ghost ids (what downloads, the other kernels and the next call's pack
// read) is enqueued lazily -- before the next thing that needs it -- so that a caller driving several ranks from
// one process can issue every rank's sweeps before any rank's wait.
int nsk_p2p_flush(nsk_graph *g) {
    if (!g || !g->p2p_close_pending) return NSK_OK;
    g->p2p_close_pending = false;
    return nsk_p2p_enqueue(g, nullptr, 0, false, 2);          // wait for tag p2p_tag + unpack
}

extern "C" {

int nsk_p2p_check(nsk_graph *g) {
    if (!g) return fail(NSK_E_INVALID, "null graph");
    if (!g->p2p_err) return NSK_OK;
    HIPCHECK(hipSetDevice(g->device));
    { int frc = nsk_p2p_flush(g); if (frc) return frc; }
    unsigned int err = 0;
    HIPCHECK(hipMemcpyAsync(&err, g->p2p_err, sizeof(err), hipMemcpyDeviceToHost, g->stream));
    HIPCHECK(hipStreamSynchronize(g->stream));
    if (err) {
        HIPCHECK(hipMemsetAsync(g->p2p_err, 0, sizeof(unsigned int), g->stream));      // reported once
        if (err & NSK_P2P_ERR_PAYLOAD)
            return fail(NSK_E_DEVICE, "peer-to-peer self-test: a peer's flag arrived but the payload read back differs from "
                                      "what the peer wrote (peer writes are not visible to this device's kernels)");
        return fail(NSK_E_DEVICE, "peer-to-peer exchange: a peer's boundary values did not arrive within "
                                  "NSK_P2P_TIMEOUT_S; the ghost values of this handle are incomplete");
    }
    return NSK_OK;
}

}  // extern "C"

// learn == 2: the memory protocol of the fused exchange (k_p2p_fused_selftest)
template <typename VT>
static int p2p_fused_selftest(nsk_graph *g, int part) {
    if (!g->p2p_peer_mask) return NSK_OK;
    if (part != 2) ++g->p2p_tag;
    P2PPlan plan;
    memset(&plan, 0, sizeof(plan));
    for (int q = 0; q < g->pworld; q++) {
        plan.base[q] = g->p2p_peer_base[q];
        plan.soff[q] = (unsigned long long)g->p_soff[q];
        plan.roff[q] = (unsigned long long)g->p_roff[q];
        plan.dbase[q] = (unsigned long long)g->p_dbase[q];
        plan.dtotal[q] = (unsigned long long)g->p_dtotal[q];
    }
    for (int q = g->pworld; q <= 16; q++) { plan.soff[q] = (unsigned long long)g->p_nsend; plan.roff[q] = (unsigned long long)g->p_nrecv; }
    k_p2p_fused_selftest<VT><<<dim3(1), dim3(NSK_BLOCK), 0, g->stream>>>((long long)g->p_nsend, (long long)g->p_nrecv, plan, g->p2p_base,
                                                                         g->pworld, g->prank, g->p2p_peer_mask, g->p2p_tag, g->p2p_err,
                                                                         g->p2p_timeout_ticks, part);
    HIPCHECK(hipGetLastError());
    return NSK_OK;
}

extern "C" {

int nsk_p2p_selftest(nsk_graph *g, int learn, int part) {
    if (!g) return fail(NSK_E_INVALID, "null graph");
    if (!g->p2p_ready) return fail(NSK_E_INVALID, "nsk_p2p_setup / nsk_p2p_export / nsk_p2p_import first");
    if (part < 0 || part > 3) return fail(NSK_E_INVALID, "bad part");
    HIPCHECK(hipSetDevice(g->device));
    { int frc = nsk_p2p_flush(g); if (frc) return frc; }
    if (learn == 2) {
        if (part == 3) return NSK_OK;
        return g->c.vbytes == 1 ? p2p_fused_selftest<int8_t>(g, part) : p2p_fused_selftest<int32_t>(g, part);
    }
    return g->c.vbytes == 1 ? p2p_exchange<int8_t>(g, nullptr, 0, learn != 0, part, 1)
                            : p2p_exchange<int32_t>(g, nullptr, 0, learn != 0, part, 1);
}

int nsk_p2p_fuse(nsk_graph *g, int on) {
    if (!g) return fail(NSK_E_INVALID, "null graph");
    if (!g->p2p_ready) return fail(NSK_E_INVALID, "nsk_p2p_setup / nsk_p2p_export / nsk_p2p_import first");
    HIPCHECK(hipSetDevice(g->device));
    { int frc = nsk_p2p_flush(g); if (frc) return frc; }
    if (on) {
        int rc = p2p_fuse_plan(g);
        if (rc) return rc;
    } else if (g->p2p_fused) {
        g->p2p_fused = false;
        g->p2p_border_tiles.clear();
        g->seg_plans_key = -1;
    }
    nsk_drop_sweep_graph(g);
    return g->p2p_fused ? 1 : 0;
}

int nsk_p2p_reset(nsk_graph *g) {
    if (!g) return fail(NSK_E_INVALID, "null graph");
    if (!g->p2p_base) return fail(NSK_E_INVALID, "nsk_p2p_setup / nsk_p2p_export first");
    HIPCHECK(hipSetDevice(g->device));
    { int frc = nsk_p2p_flush(g); if (frc) return frc; }
    HIPCHECK(hipStreamSynchronize(g->stream));
    nsk_drop_sweep_graph(g);
    HIPCHECK(hipMemsetAsync(g->p2p_base, 0, g->p2p_bytes, g->stream));
    HIPCHECK(hipMemsetAsync(g->p2p_err, 0, 4 * sizeof(unsigned int), g->stream));
    HIPCHECK(hipStreamSynchronize(g->stream));
    g->p2p_tag = 0;
    g->p2p_close_pending = false;
    return NSK_OK;
}

int nsk_p2p_exchange(nsk_graph *g, int learn, int part) {
    if (!g) return fail(NSK_E_INVALID, "null graph");
    if (!g->p2p_ready) return fail(NSK_E_INVALID, "nsk_p2p_setup / nsk_p2p_export / nsk_p2p_import first");
    if (part < 0 || part > 3) return fail(NSK_E_INVALID, "bad part");
    HIPCHECK(hipSetDevice(g->device));
    { int frc = nsk_p2p_flush(g); if (frc) return frc; }
    return nsk_p2p_enqueue(g, nullptr, 0, learn != 0, part);
}

int nsk_gibbs_sweeps_p2p(nsk_graph *g, int64_t nsweeps, int sample_evidence, int burnin) {
    if (!g) return fail(NSK_E_INVALID, "null graph");
    if (!g->p2p_ready) return fail(NSK_E_INVALID, "nsk_p2p_setup / nsk_p2p_export / nsk_p2p_import first");
    if (nsweeps < 0 || nsweeps > INT32_MAX) return fail(NSK_E_INVALID, "bad sweep count");
    HIPCHECK(hipSetDevice(g->device));
    return nsk_gibbs_run(g, nsweeps, sample_evidence, burnin, true);      // (flushes a pending close unless it continues it)
}

int nsk_learn_sweeps_p2p(nsk_graph *g, int64_t nsweeps, double step, double decay, int regularization,
                         double reg_param, int64_t truncation, int learn_non_evidence) {
    if (!g) return fail(NSK_E_INVALID, "null graph");
    if (!g->p2p_ready) return fail(NSK_E_INVALID, "nsk_p2p_setup / nsk_p2p_export / nsk_p2p_import first");
    if (nsweeps < 0 || nsweeps > INT32_MAX) return fail(NSK_E_INVALID, "bad sweep count");
    HIPCHECK(hipSetDevice(g->device));
    { int frc = nsk_p2p_flush(g); if (frc) return frc; }
    const int nw = (int)g->c.nweight;
    // the caller may have written the weight buffer since the last epoch: this call starts from what is there
    if (nw && nsweeps) HIPCHECK(hipMemcpyAsync(g->w_start, g->w, (size_t)nw * sizeof(double), hipMemcpyDeviceToDevice, g->stream));
    for (int64_t s = 0; s < nsweeps; s++) {
        int rc = nsk_learn_sweeps(g, 1, step, 1.0, regularization, reg_param, truncation, learn_non_evidence);
        if (rc) return rc;
        if ((rc = nsk_p2p_enqueue(g, nullptr, 0, true, 0))) return rc;     // values of both chains + weight deltas; w_start = merged w
        step *= decay;
    }
    return NSK_OK;
}

// ---- native RCCL loop -----------------------------------------------------------------------------
static int load_rccl(const char *path) {
    if (g_rccl.lib) return NSK_OK;
    void *h = dlopen(path && path[0] ? path : "librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return fail(NSK_E_DEVICE, std::string("dlopen(librccl): ") + dlerror());
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
    g_rccl.AllGather = (decltype(g_rccl.AllGather))dlsym(h, "ncclAllGather");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(h, "ncclAllReduce");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllGather || !g_rccl.AllReduce || !g_rccl.CommDestroy)
        return fail(NSK_E_DEVICE, "librccl lacks the expected symbols");
    g_rccl.lib = h;
    return NSK_OK;
}

#define RCCLCHECK(expr)                                                                         \
    do {                                                                                        \
        ncclResult_t r_ = (expr);                                                               \
        if (r_ != ncclSuccess)                                                                  \
            return fail(NSK_E_DEVICE, std::string(#expr) + ": " +                              \
                        (g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "rccl error"));    \
    } while (0)

int nsk_comm_unique_id(const char *librccl_path, void *id128) {
    if (!id128) return fail(NSK_E_INVALID, "null argument");
    int rc = load_rccl(librccl_path);
    if (rc) return rc;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    RCCLCHECK(g_rccl.GetUniqueId((ncclUniqueId *)id128));
    return NSK_OK;
}

int nsk_comm_init(nsk_graph *g, int world, int rank, const void *id128, const char *librccl_path) {
    if (!g || !id128 || world < 1 || rank < 0 || rank >= world) return fail(NSK_E_INVALID, "bad argument");
    int rc = load_rccl(librccl_path);
    if (rc) return rc;
    HIPCHECK(hipSetDevice(g->device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t comm = nullptr;
    RCCLCHECK(g_rccl.CommInitRank(&comm, world, id, rank));
    g->rccl_comm = comm;
    return NSK_OK;
}

static int native_exchange(nsk_graph *g, int which) {
    int rc = exchange_step(g, which, true);
    if (rc) return rc;
    const void *sb = which == NSK_BUF_VALUE ? g->x_send : g->x_send_evid;
    void *rb = which == NSK_BUF_VALUE ? g->x_recv : g->x_recv_evid;
    if (g->xslot > 0)
        RCCLCHECK(g_rccl.AllGather(sb, rb, (size_t)g->xslot, g->c.vbytes == 1 ? ncclInt8 : ncclInt32,
                                   (ncclComm_t)g->rccl_comm, g->stream));
    return exchange_step(g, which, false);
}

int nsk_gibbs_sweeps_exchange(nsk_graph *g, int64_t nsweeps, int sample_evidence, int burnin) {
    if (!g) return fail(NSK_E_INVALID, "null graph");
    if (!g->rccl_comm || g->xworld == 0) return fail(NSK_E_INVALID, "nsk_exchange_setup / nsk_comm_init first");
    for (int64_t s = 0; s < nsweeps; s++) {
        int rc = nsk_gibbs_sweeps(g, 1, sample_evidence, burnin);
        if (rc) return rc;
        if ((rc = native_exchange(g, NSK_BUF_VALUE))) return rc;
    }
    return NSK_OK;
}

int nsk_learn_sweeps_exchange(nsk_graph *g, int64_t nsweeps, double step, double decay, int regularization,
                              double reg_param, int64_t truncation, int learn_non_evidence) {
    if (!g) return fail(NSK_E_INVALID, "null graph");
    if (!g->rccl_comm || g->xworld == 0) return fail(NSK_E_INVALID, "nsk_exchange_setup / nsk_comm_init first");
    const int nw = (int)g->c.nweight;
    for (int64_t s = 0; s < nsweeps; s++) {
        HIPCHECK(hipSetDevice(g->device));
        if (nw) HIPCHECK(hipMemcpyAsync(g->w_start, g->w, (size_t)nw * sizeof(double), hipMemcpyDeviceToDevice, g->stream));
        int rc = nsk_learn_sweeps(g, 1, step, 1.0, regularization, reg_param, truncation, learn_non_evidence);
        if (rc) return rc;
        if ((rc = native_exchange(g, NSK_BUF_VALUE))) return rc;
        if ((rc = native_exchange(g, NSK_BUF_VALUE_EVID))) return rc;
        if (nw) {       // w = w_start + sum over ranks of (w - w_start): numbskull_master.py:223-224
            const dim3 grid((nw + NSK_BLOCK - 1) / NSK_BLOCK), block(NSK_BLOCK);
            k_weight_delta<<<grid, block, 0, g->stream>>>(g->w, g->w_start, g->w_delta, nw);
            RCCLCHECK(g_rccl.AllReduce(g->w_delta, g->w_delta, (size_t)nw, ncclDouble, ncclSum,
                                       (ncclComm_t)g->rccl_comm, g->stream));
            k_weight_merge<<<grid, block, 0, g->stream>>>(g->w, g->w_start, g->w_delta, nw);
            g->weights_dirty = true;
        }
        step *= decay;
    }
    return NSK_OK;
}

}  // extern "C"
