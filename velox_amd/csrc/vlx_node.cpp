// One node, many GPUs (include/velox_amd.h): the exchange steps of a sharded encode, the decode partitioning,
// and the in-process pool behind the host-pointer drop-ins.
//
// What ProcessBlock computes over the whole input is the only thing exchanged between ranks: the file-global
// high_res flag (VeloxCore.h:232-236) and the in-order join of the sub-block futures (VeloxCore.h:299-303) = the
// table of chunk byte sizes.  The collectives sit behind one small interface (Coll) with three transports:
//   RcclColl    one process per GPU, RCCL over xGMI (librccl.so.1 bound at run time)
//   LocalColl   the ranks are threads of THIS process, one vlx_ctx each (any devices, the same one several times
//               included): a generation barrier on the host and device-to-device copies.  This is what the pool
//               (vlx_pool_*: ProcessBlock over all GPUs from the reference's single caller thread) runs on.
//   CustomColl  function pointers of the caller (MPI, pipes of a test harness)
// vlx_node_* above them is the same code for all three.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <dlfcn.h>
#include <rccl/rccl.h>

#include "vlx_host.h"

hipError_t vlx_launch_node_row(hipStream_t st, const ChunkInfo *ci, uint32_t count, uint32_t head, const uint32_t *d_flag, uint32_t *row, uint32_t bound);

namespace {

// ---- the transports -------------------------------------------------------------------------------
struct Coll {
    virtual ~Coll() {}
    // every buffer is DEVICE memory of the calling rank's GPU; every call is made by all ranks in the same order
    virtual int all_reduce_max_u32(uint32_t *d, size_t n, hipStream_t st, std::string &err) = 0;
    virtual int all_gather(const void *d_send, void *d_recv, size_t bytes, hipStream_t st, std::string &err) = 0;
    virtual int group_start(std::string &err) = 0;
    virtual int send(const void *d, size_t bytes, int peer, hipStream_t st, std::string &err) = 0;
    virtual int recv(void *d, size_t bytes, int peer, hipStream_t st, std::string &err) = 0;
    virtual int group_end(hipStream_t st, std::string &err) = 0;
};

struct Rccl {
    void *lib = nullptr; bool ok = false;
    decltype(&ncclGetUniqueId) GetUniqueId; decltype(&ncclCommInitRank) CommInitRank; decltype(&ncclCommDestroy) CommDestroy;
    decltype(&ncclAllReduce) AllReduce; decltype(&ncclAllGather) AllGather; decltype(&ncclSend) Send; decltype(&ncclRecv) Recv;
    decltype(&ncclGroupStart) GroupStart; decltype(&ncclGroupEnd) GroupEnd; decltype(&ncclGetErrorString) GetErrorString;
    Rccl()
    {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        lib = dlopen(names[0], RTLD_NOW | RTLD_NOLOAD);                   // already in the process?
        for (int i = 0; !lib && i < 3; i++) lib = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
        if (!lib) return;
#define VLX_SYM(f) f = (decltype(f))dlsym(lib, "nccl" #f); if (!f) return
        VLX_SYM(GetUniqueId); VLX_SYM(CommInitRank); VLX_SYM(CommDestroy); VLX_SYM(AllReduce); VLX_SYM(AllGather);
        VLX_SYM(Send); VLX_SYM(Recv); VLX_SYM(GroupStart); VLX_SYM(GroupEnd); VLX_SYM(GetErrorString);
#undef VLX_SYM
        ok = true;
    }
};
Rccl *rccl() { static Rccl r; return r.ok ? &r : nullptr; }

struct RcclColl : Coll {
    ncclComm_t comm = nullptr;
    ~RcclColl() override { if (comm && rccl()) (void)rccl()->CommDestroy(comm); }
    int chk(ncclResult_t r, const char *what, std::string &err)
    {
        if (r == ncclSuccess) return VLX_OK;
        err = std::string(what) + ": " + rccl()->GetErrorString(r);
        return VLX_ERR_HIP;
    }
    int all_reduce_max_u32(uint32_t *d, size_t n, hipStream_t st, std::string &err) override
    { return chk(rccl()->AllReduce(d, d, n, ncclUint32, ncclMax, comm, st), "ncclAllReduce", err); }
    int all_gather(const void *s, void *r, size_t bytes, hipStream_t st, std::string &err) override
    { return chk(rccl()->AllGather(s, r, bytes, ncclUint8, comm, st), "ncclAllGather", err); }
    int group_start(std::string &err) override { return chk(rccl()->GroupStart(), "ncclGroupStart", err); }
    int send(const void *d, size_t bytes, int peer, hipStream_t st, std::string &err) override
    { return chk(rccl()->Send(d, bytes, ncclUint8, peer, comm, st), "ncclSend", err); }
    int recv(void *d, size_t bytes, int peer, hipStream_t st, std::string &err) override
    { return chk(rccl()->Recv(d, bytes, ncclUint8, peer, comm, st), "ncclRecv", err); }
    int group_end(hipStream_t, std::string &err) override { return chk(rccl()->GroupEnd(), "ncclGroupEnd", err); }
};

struct CustomColl : Coll {
    vlx_collectives f;
    int chk(int rc, const char *what, std::string &err) { if (rc == 0) return VLX_OK; err = std::string(what) + " (caller's transport) failed"; return VLX_ERR_HIP; }
    int all_reduce_max_u32(uint32_t *d, size_t n, hipStream_t st, std::string &err) override { return chk(f.all_reduce_max_u32(f.user, d, n, st), "all_reduce_max_u32", err); }
    int all_gather(const void *s, void *r, size_t bytes, hipStream_t st, std::string &err) override { return chk(f.all_gather(f.user, s, r, bytes, st), "all_gather", err); }
    int group_start(std::string &err) override { return f.group_start ? chk(f.group_start(f.user), "group_start", err) : VLX_OK; }
    int send(const void *d, size_t bytes, int peer, hipStream_t st, std::string &err) override { return chk(f.send(f.user, d, bytes, peer, st), "send", err); }
    int recv(void *d, size_t bytes, int peer, hipStream_t st, std::string &err) override { return chk(f.recv(f.user, d, bytes, peer, st), "recv", err); }
    int group_end(hipStream_t st, std::string &err) override { return f.group_end ? chk(f.group_end(f.user, st), "group_end", err) : VLX_OK; }
};

} // namespace

// Ranks that are threads of one process.  Every collective is: make my buffers known, meet at the barrier, copy
// what I need out of my peers' buffers (device to device, or through the host for the few flag words), meet again
// so that nobody reuses a buffer a peer is still reading.  A rank that does not show up within the time limit
// breaks the group: every later call fails instead of hanging.
struct vlx_local_group {
    int world = 1;
    std::mutex m; std::condition_variable cv;
    int arrived = 0; uint64_t gen = 0; bool broken = false;
    int timeout_s = 120;
    struct P2P { const void *src; size_t bytes; int peer; bool taken; };
    struct Slot { const void *send = nullptr; size_t bytes = 0; std::vector<P2P> sends; };
    std::vector<Slot> slot;
    std::atomic<int> refs{1};

    bool barrier()
    {
        std::unique_lock<std::mutex> lk(m);
        if (broken) return false;
        const uint64_t g = gen;
        if (++arrived == world) { arrived = 0; gen++; cv.notify_all(); return true; }
        if (!cv.wait_for(lk, std::chrono::seconds(timeout_s), [&] { return gen != g || broken; })) { broken = true; cv.notify_all(); return false; }
        return !broken;
    }
};

namespace {

struct LocalColl : Coll {
    vlx_local_group *g; int rank;
    std::vector<vlx_local_group::P2P> recvs; std::vector<vlx_local_group::P2P> my_sends; std::vector<void *> recv_dst;
    LocalColl(vlx_local_group *grp, int r) : g(grp), rank(r) { g->refs++; }
    ~LocalColl() override { if (--g->refs == 0) delete g; }
    static int hip(hipError_t e, const char *what, std::string &err) { if (e == hipSuccess) return VLX_OK; err = std::string(what) + ": " + hipGetErrorString(e); return VLX_ERR_HIP; }
    int meet(std::string &err) { if (g->barrier()) return VLX_OK; err = "a rank of the local group did not reach the collective (group broken)"; return VLX_ERR_PEER; }

    int all_reduce_max_u32(uint32_t *d, size_t n, hipStream_t st, std::string &err) override
    {
        int rc = hip(hipStreamSynchronize(st), "hipStreamSynchronize", err);   // my words are final
        g->slot[(size_t)rank].send = d; g->slot[(size_t)rank].bytes = 4 * n;
        { const int b = meet(err); if (b != VLX_OK) return b; }
        std::vector<uint32_t> acc(n, 0), tmp(n);
        for (int p = 0; p < g->world && rc == VLX_OK; p++) {
            if (g->slot[(size_t)p].bytes != 4 * n) { err = "all_reduce: ranks disagree on the word count"; rc = VLX_ERR_ARG; break; }
            rc = hip(hipMemcpy(tmp.data(), g->slot[(size_t)p].send, 4 * n, hipMemcpyDeviceToHost), "hipMemcpy(flag words)", err);
            for (size_t i = 0; i < n; i++) acc[i] = std::max(acc[i], tmp[i]);
        }
        { const int b = meet(err); if (b != VLX_OK) return b; }                 // everybody has read everybody
        if (rc == VLX_OK) rc = hip(hipMemcpy(d, acc.data(), 4 * n, hipMemcpyHostToDevice), "hipMemcpy(reduced words)", err);
        return rc;
    }
    int all_gather(const void *s, void *r, size_t bytes, hipStream_t st, std::string &err) override
    {
        int rc = hip(hipStreamSynchronize(st), "hipStreamSynchronize", err);
        g->slot[(size_t)rank].send = s; g->slot[(size_t)rank].bytes = bytes;
        { const int b = meet(err); if (b != VLX_OK) return b; }
        for (int p = 0; p < g->world && rc == VLX_OK; p++) {
            if (g->slot[(size_t)p].bytes != bytes) { err = "all_gather: ranks disagree on the row length"; rc = VLX_ERR_ARG; break; }
            if (bytes) rc = hip(hipMemcpyAsync((uint8_t *)r + (size_t)p * bytes, g->slot[(size_t)p].send, bytes, hipMemcpyDefault, st), "hipMemcpyAsync(gather)", err);
        }
        if (rc == VLX_OK) rc = hip(hipStreamSynchronize(st), "hipStreamSynchronize", err);
        { const int b = meet(err); if (b != VLX_OK) return b; }
        return rc;
    }
    int group_start(std::string &) override { recvs.clear(); recv_dst.clear(); my_sends.clear(); return VLX_OK; }
    int send(const void *d, size_t bytes, int peer, hipStream_t, std::string &) override { my_sends.push_back({d, bytes, peer, false}); return VLX_OK; }
    int recv(void *d, size_t bytes, int peer, hipStream_t, std::string &) override { recvs.push_back({nullptr, bytes, peer, false}); recv_dst.push_back(d); return VLX_OK; }
    int group_end(hipStream_t st, std::string &err) override
    {
        int rc = hip(hipStreamSynchronize(st), "hipStreamSynchronize", err);
        g->slot[(size_t)rank].sends = my_sends;
        { const int b = meet(err); if (b != VLX_OK) return b; }
        std::vector<size_t> next((size_t)g->world, 0);                      // the k-th receive from a peer takes its k-th send to me
        for (size_t i = 0; i < recvs.size() && rc == VLX_OK; i++) {
            const int p = recvs[i].peer;
            const std::vector<vlx_local_group::P2P> &ps = g->slot[(size_t)p].sends;
            size_t &k = next[(size_t)p];
            while (k < ps.size() && ps[k].peer != rank) k++;
            if (k >= ps.size() || ps[k].bytes != recvs[i].bytes) { err = "recv without a matching send of the same size"; rc = VLX_ERR_ARG; break; }
            if (recvs[i].bytes) rc = hip(hipMemcpyAsync(recv_dst[i], ps[k].src, recvs[i].bytes, hipMemcpyDefault, st), "hipMemcpyAsync(p2p)", err);
            k++;
        }
        if (rc == VLX_OK) rc = hip(hipStreamSynchronize(st), "hipStreamSynchronize", err);
        { const int b = meet(err); if (b != VLX_OK) return b; }
        return rc;
    }
};

} // namespace

// ---- the node -------------------------------------------------------------------------------------
#define NODE_ROW_FAILED 0xFFFFFFFFu                 // row[0] of a rank that is about to return an error
#define NODE_FLAG_WORDS VLX_HIRES_STRIDE            // the all-reduced line: [high_res, chunk count, failed, odd total, ...]

struct vlx_node {
    vlx_ctx *ctx = nullptr; Coll *coll = nullptr; int rank = 0, world = 1;
    DevBuf d_row, d_table, d_seam, d_flags;        // collective staging in HBM
    HostBuf h_row, h_table, h_flags;               // pinned mirrors
    std::vector<std::vector<uint32_t>> sizes;      // [rank] chunk byte sizes of the last vlx_node_encode_track
    std::vector<uint64_t> off, seg_len; uint64_t total_len = 0;
    // state of the encode_track call in flight (the hooks run inside vlx_encode_device_impl)
    uint32_t mine = 0, bound = 0; bool bound_from_peers = false, did_flags = false, did_rows = false;
};

#define NCHK(node, expr) do { const int rc__ = (expr); if (rc__ != VLX_OK) return rc__; } while (0)

// The all-reduce of one flag line.  Word 0 is the shard's high_res flag (k_hires_scan wrote it), words 1-3 are this
// rank's chunk count, "I am about to fail" and "my sample count is odd"; MAX over ranks of each.  With no explicit
// row bound the reduced chunk count IS the bound, which costs one synchronisation here.
static int node_flags(vlx_node *n, uint32_t *d_flags, uint32_t failed, uint32_t odd, hipStream_t st)
{
    vlx_ctx *c = n->ctx;
    HIPCHK(c, n->h_flags.ensure(sizeof(uint32_t) * 8));
    uint32_t *hf = (uint32_t *)n->h_flags.p;
    hf[0] = n->mine; hf[1] = failed; hf[2] = odd;
    HIPCHK(c, hipMemcpyAsync(d_flags + 1, hf, sizeof(uint32_t) * 3, hipMemcpyHostToDevice, st));
    n->did_flags = true;
    NCHK(n, n->coll->all_reduce_max_u32(d_flags, NODE_FLAG_WORDS, st, c->err));
    if (n->bound_from_peers || failed) {
        HIPCHK(c, hipMemcpyAsync(hf + 4, d_flags, sizeof(uint32_t) * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        if (n->bound_from_peers) n->bound = std::max<uint32_t>(hf[5], 1);
    }
    return VLX_OK;
}
static int node_flag_hook(void *self, uint32_t *d_flags, size_t, hipStream_t st) { return node_flags((vlx_node *)self, d_flags, 0, 0, st); }

// The all-gather of [count, high_res, sizes...] rows, filled on the device straight from k_chunk_layout's records and
// brought to pinned memory behind it; nothing waits here (the caller's one synchronisation follows k_assemble).
static int node_rows(vlx_node *n, const ChunkInfo *d_ci, uint64_t n_chunks, int status, const uint32_t *d_flag, hipStream_t st)
{
    vlx_ctx *c = n->ctx;
    const size_t row = 2 + (size_t)n->bound;
    if (status == VLX_OK && n_chunks > n->bound) { c->err = "more chunks on this rank than max_chunks_per_rank"; status = VLX_ERR_ARG; }
    HIPCHK(c, n->d_row.ensure(sizeof(uint32_t) * row)); HIPCHK(c, n->d_table.ensure(sizeof(uint32_t) * row * (size_t)n->world));
    HIPCHK(c, n->h_table.ensure(sizeof(uint32_t) * row * (size_t)n->world));
    HIPCHK(c, vlx_launch_node_row(st, d_ci, status == VLX_OK ? (uint32_t)n_chunks : 0u, status == VLX_OK ? (uint32_t)n_chunks : NODE_ROW_FAILED,
                                  d_flag, (uint32_t *)n->d_row.p, n->bound));
    n->did_rows = true;
    NCHK(n, n->coll->all_gather(n->d_row.p, n->d_table.p, sizeof(uint32_t) * row, st, c->err));
    HIPCHK(c, hipMemcpyAsync(n->h_table.p, n->d_table.p, sizeof(uint32_t) * row * (size_t)n->world, hipMemcpyDeviceToHost, st));
    return status;
}
static int node_rows_hook(void *self, const ChunkInfo *d_ci, uint64_t n_chunks, int status, hipStream_t st)
{
    vlx_node *n = (vlx_node *)self;
    return node_rows(n, d_ci, n_chunks, status, (const uint32_t *)n->ctx->track_hires.p, st);
}

// A rank that cannot go on still makes the calls its peers are waiting in, and says so in them.
static int node_fail_collectively(vlx_node *n, int rc)
{
    vlx_ctx *c = n->ctx;
    const std::string why = c->err;
    hipStream_t st = c->stream;
    if (hipSetDevice(c->device) == hipSuccess && n->d_flags.ensure(sizeof(uint32_t) * NODE_FLAG_WORDS) == hipSuccess) {
        uint32_t *df = (uint32_t *)n->d_flags.p;
        if (!n->did_flags) {
            (void)hipMemsetAsync(df, 0, sizeof(uint32_t) * NODE_FLAG_WORDS, st);
            n->mine = 0;
            (void)node_flags(n, df, 1, 0, st);
        }
        if (!n->did_rows) {
            (void)hipMemsetAsync(df, 0, sizeof(uint32_t), st);
            (void)node_rows(n, nullptr, 0, rc, df, st);
        }
        (void)hipStreamSynchronize(st);
    }
    c->err = why;
    return rc;
}

extern "C" {

int vlx_node_unique_id(uint8_t id[VLX_NODE_ID_BYTES])
{
    if (!id) return VLX_ERR_ARG;
    Rccl *R = rccl();
    if (!R) return VLX_ERR_UNSUPPORTED;
    static_assert(sizeof(ncclUniqueId) == VLX_NODE_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId u;
    if (R->GetUniqueId(&u) != ncclSuccess) return VLX_ERR_HIP;
    memcpy(id, &u, sizeof(u));
    return VLX_OK;
}

static vlx_node *node_new(vlx_ctx *c, Coll *coll, int rank, int world)
{
    vlx_node *n = new (std::nothrow) vlx_node();
    if (!n) { delete coll; return nullptr; }
    n->ctx = c; n->coll = coll; n->rank = rank; n->world = world;
    n->sizes.resize((size_t)world); n->off.assign((size_t)world, 0); n->seg_len.assign((size_t)world, 0);
    return n;
}

int vlx_node_init(vlx_ctx *c, const uint8_t id[VLX_NODE_ID_BYTES], int rank, int world, vlx_node **out)
{
    if (!c || !id || !out || world < 1 || rank < 0 || rank >= world) return VLX_ERR_ARG;
    *out = nullptr;
    Rccl *R = rccl();
    if (!R) { c->err = "librccl.so.1 not found: the multi-GPU exchange needs RCCL"; return VLX_ERR_UNSUPPORTED; }
    HIPCHK(c, hipSetDevice(c->device));
    RcclColl *coll = new (std::nothrow) RcclColl();
    if (!coll) return VLX_ERR_NOMEM;
    ncclUniqueId u; memcpy(&u, id, sizeof(u));
    const ncclResult_t r = R->CommInitRank(&coll->comm, world, u, rank);
    if (r != ncclSuccess) { c->err = std::string("ncclCommInitRank: ") + R->GetErrorString(r); coll->comm = nullptr; delete coll; return VLX_ERR_HIP; }
    *out = node_new(c, coll, rank, world);
    return *out ? VLX_OK : VLX_ERR_NOMEM;
}

int vlx_local_group_create(int world, vlx_local_group **out)
{
    if (!out || world < 1) return VLX_ERR_ARG;
    vlx_local_group *g = new (std::nothrow) vlx_local_group();
    if (!g) return VLX_ERR_NOMEM;
    g->world = world; g->slot.resize((size_t)world);
    *out = g;
    return VLX_OK;
}
void vlx_local_group_destroy(vlx_local_group *g) { if (g && --g->refs == 0) delete g; }
int vlx_local_group_set_timeout(vlx_local_group *g, int seconds) { if (!g || seconds < 1) return VLX_ERR_ARG; g->timeout_s = seconds; return VLX_OK; }

int vlx_node_init_local(vlx_ctx *c, vlx_local_group *g, int rank, vlx_node **out)
{
    if (!c || !g || !out || rank < 0 || rank >= g->world) return VLX_ERR_ARG;
    LocalColl *coll = new (std::nothrow) LocalColl(g, rank);
    if (!coll) return VLX_ERR_NOMEM;
    *out = node_new(c, coll, rank, g->world);
    return *out ? VLX_OK : VLX_ERR_NOMEM;
}

int vlx_node_init_custom(vlx_ctx *c, const vlx_collectives *fns, int rank, int world, vlx_node **out)
{
    if (!c || !fns || !out || world < 1 || rank < 0 || rank >= world) return VLX_ERR_ARG;
    if (!fns->all_reduce_max_u32 || !fns->all_gather || !fns->send || !fns->recv) return VLX_ERR_ARG;
    CustomColl *coll = new (std::nothrow) CustomColl();
    if (!coll) return VLX_ERR_NOMEM;
    coll->f = *fns;
    *out = node_new(c, coll, rank, world);
    return *out ? VLX_OK : VLX_ERR_NOMEM;
}

void vlx_node_destroy(vlx_node *n)
{
    if (!n) return;
    (void)hipSetDevice(n->ctx->device);
    (void)hipStreamSynchronize(n->ctx->stream);
    delete n->coll;
    n->d_row.release(); n->d_table.release(); n->d_seam.release(); n->d_flags.release();
    n->h_row.release(); n->h_table.release(); n->h_flags.release();
    delete n;
}

void vlx_node_shard_range(uint64_t total_samples, int rank, int world, uint64_t *first_sample, uint64_t *n_samples)
{
    const uint64_t n_chunks = vlx_track_chunks(total_samples);
    const uint64_t base = n_chunks / (uint64_t)world, extra = n_chunks % (uint64_t)world;
    const uint64_t c0 = (uint64_t)rank * base + std::min<uint64_t>((uint64_t)rank, extra);
    const uint64_t nc = base + ((uint64_t)rank < extra ? 1 : 0);
    const uint64_t s0 = std::min(c0 * VLX_SUB_BLOCK, total_samples), s1 = std::min((c0 + nc) * VLX_SUB_BLOCK, total_samples);
    if (first_sample) *first_sample = s0;
    if (n_samples) *n_samples = s1 - s0;
}

int vlx_node_offset_table(const uint32_t *counts, const uint32_t *size_rows, uint32_t world, uint32_t row_stride,
                          uint64_t *off, uint64_t *seg_len, uint64_t *total_len)
{
    if (!counts || (!size_rows && row_stride) || !world) return VLX_ERR_ARG;
    uint64_t pos = 0;
    for (uint32_t r = 0; r < world; r++) {
        if (counts[r] > row_stride) return VLX_ERR_ARG;
        uint64_t body = 0;
        for (uint32_t i = 0; i < counts[r]; i++) body += (uint64_t)size_rows[(size_t)r * row_stride + i] + 4;   // u32 prefix + bytes (VeloxCore.h:301-302)
        if (off) off[r] = pos;
        if (seg_len) seg_len[r] = body + 1;                               // the 2 header bits make a shard span one byte more than its body
        pos += body;
    }
    if (total_len) *total_len = pos + 1;
    return VLX_OK;
}

int vlx_node_encode_track(vlx_node *n, const void *d_pcm, int fmt, uint64_t n_samples, uint32_t max_chunks_per_rank,
                          uint8_t *d_payload, size_t payload_cap, uint64_t *shard_len, uint64_t *shard_off,
                          uint64_t *total_len, int *high_res)
{
    if (!n) return VLX_ERR_ARG;
    vlx_ctx *c = n->ctx;
    n->did_flags = n->did_rows = false;
    n->mine = (uint32_t)vlx_track_chunks(n_samples);
    n->bound_from_peers = max_chunks_per_rank == 0; n->bound = max_chunks_per_rank;
    for (auto &v : n->sizes) v.clear();
    // ---- everything that can be refused is refused BEFORE the first collective, and refused collectively
    int pre = VLX_OK;
    if (!shard_len || (n_samples && (!d_pcm || !d_payload))) { c->err = "null argument"; pre = VLX_ERR_ARG; }
    else if (fmt != VLX_FMT_S16 && fmt != VLX_FMT_S24 && fmt != VLX_FMT_S32 && fmt != VLX_FMT_I64) { c->err = "bad sample format"; pre = VLX_ERR_ARG; }
    else if ((uintptr_t)d_payload & 15) { c->err = "d_payload must be 16-byte aligned"; pre = VLX_ERR_ARG; }
    else if (vlx_track_chunks(n_samples) >= (1ull << 31)) { c->err = "too many chunks in one batch"; pre = VLX_ERR_UNSUPPORTED; }
    else if (!n->bound_from_peers && n->mine > n->bound) { c->err = "more chunks on this rank than max_chunks_per_rank"; pre = VLX_ERR_ARG; }
    else if ((n_samples & 1) && n->world > 1) {
        // an odd total is ONE stream over the whole input (VeloxCore.h:242-253): there is nothing to cut
        c->err = "an odd sample total is a single stream (VeloxCore.h:242-253) and cannot be sharded: encode it on one GPU (vlx_encode_device)";
        pre = VLX_ERR_UNSUPPORTED;
    }
    if (pre != VLX_OK) return node_fail_collectively(n, pre);
    HIPCHK(c, hipSetDevice(c->device));
    if (n_samples & 1) {                                                  // a world of one rank: the odd path, as vlx_encode_device routes it
        const uint64_t off0 = 0; uint64_t poff = 0;
        const int rc = vlx_encode_device(c, d_pcm, fmt, &off0, &n_samples, 1, d_payload, payload_cap, &poff, shard_len);
        if (rc != VLX_OK) return rc;
        if (poff) HIPCHK(c, hipMemcpy(d_payload, d_payload + poff, *shard_len, hipMemcpyDeviceToDevice));
        n->off[0] = 0; n->seg_len[0] = *shard_len; n->total_len = *shard_len;
        if (shard_off) *shard_off = 0;
        if (total_len) *total_len = *shard_len;
        if (high_res) *high_res = *shard_len ? -1 : 0;                    // the bit sits in the payload; no table entry for an odd track
        uint8_t first = 0;
        if (*shard_len) HIPCHK(c, hipMemcpy(&first, d_payload, 1, hipMemcpyDeviceToHost));
        if (high_res) *high_res = (first >> 1) & 1;
        n->sizes[0].clear();
        return VLX_OK;
    }
    const uint64_t off0 = 0; uint64_t poff = 0;
    const int force = n->rank == 0 ? -1 : -3;                             // scan on the device; ranks > 0 hold continuation shards
    EncHook hook{node_flag_hook, node_rows_hook, n};
    int rc = vlx_encode_device_impl(c, d_pcm, fmt, &off0, &n_samples, 1, nullptr, nullptr, &force, d_payload, payload_cap, &poff, shard_len, &hook);
    if (rc != VLX_OK && (!n->did_flags || !n->did_rows)) return node_fail_collectively(n, rc);
    // ---- the table came down behind k_assemble (or behind the failed rank's row): one row per rank
    const size_t row = 2 + (size_t)n->bound;
    const uint32_t *tab = (const uint32_t *)n->h_table.p;
    std::vector<uint32_t> counts((size_t)n->world);
    bool peer_failed = false;
    for (int r = 0; r < n->world; r++) {
        const uint32_t cnt = tab[(size_t)r * row];
        if (cnt == NODE_ROW_FAILED) { peer_failed = true; counts[(size_t)r] = 0; continue; }
        if (cnt > n->bound) { c->err = "a rank sent more chunks than max_chunks_per_rank"; return VLX_ERR_ARG; }
        counts[(size_t)r] = cnt;
        n->sizes[(size_t)r].assign(tab + (size_t)r * row + 2, tab + (size_t)r * row + 2 + cnt);
    }
    if (rc != VLX_OK) return rc;
    if (peer_failed) { c->err = "another rank failed its share of the track"; return VLX_ERR_PEER; }
    std::vector<uint32_t> rows((size_t)n->world * std::max<uint32_t>(n->bound, 1), 0);
    for (int r = 0; r < n->world; r++) std::copy(n->sizes[(size_t)r].begin(), n->sizes[(size_t)r].end(), rows.begin() + (size_t)r * n->bound);
    rc = vlx_node_offset_table(counts.data(), rows.data(), (uint32_t)n->world, n->bound, n->off.data(), n->seg_len.data(), &n->total_len);
    if (rc != VLX_OK) return rc;
    if (*shard_len != n->seg_len[(size_t)n->rank]) { c->err = "shard payload length disagrees with its chunk sizes"; return VLX_ERR_HIP; }
    if (shard_off) *shard_off = n->off[(size_t)n->rank];
    if (total_len) *total_len = n->total_len;
    if (high_res) *high_res = (fmt == VLX_FMT_S16) ? 0 : (tab[(size_t)n->rank * row + 1] ? 1 : 0);
    return VLX_OK;
}

int vlx_node_chunk_sizes(const vlx_node *n, int rank, uint32_t *h_sizes, size_t cap, size_t *n_chunks)
{
    if (!n || !n_chunks || rank < 0 || rank >= n->world) return VLX_ERR_ARG;
    const std::vector<uint32_t> &v = n->sizes[(size_t)rank];
    *n_chunks = v.size();
    if (!h_sizes) return VLX_OK;
    if (cap < v.size()) return VLX_ERR_CAPACITY;
    std::copy(v.begin(), v.end(), h_sizes);
    return VLX_OK;
}

int vlx_node_gather_payload(vlx_node *n, int root, const uint8_t *d_shard, uint64_t shard_len, uint8_t *d_full, size_t full_cap)
{
    if (!n) return VLX_ERR_ARG;
    vlx_ctx *c = n->ctx;
    // a rank whose arguments are bad still takes part in the first exchange and says so there: nobody is left waiting
    int pre = VLX_OK;
    if (root < 0 || root >= n->world || (shard_len && !d_shard)) { c->err = "bad root / null shard"; pre = VLX_ERR_ARG; }
    else if (shard_len != n->seg_len[(size_t)n->rank]) { c->err = "shard length differs from the last vlx_node_encode_track"; pre = VLX_ERR_ARG; }
    const bool is_root = pre == VLX_OK && n->rank == root;
    if (is_root && (!d_full || full_cap < n->total_len)) { c->err = "root buffer too small for the whole payload"; pre = VLX_ERR_CAPACITY; }
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = c->stream;
    // every shard's first byte (the seam with its predecessor) + a "failed" bit, all-gathered as one u32 per rank
    HIPCHK(c, n->d_seam.ensure(sizeof(uint32_t) * (1 + (size_t)n->world) + sizeof(uint64_t) * (size_t)n->world + 64));
    HIPCHK(c, n->h_row.ensure(sizeof(uint64_t) * ((size_t)n->world + 2) + sizeof(uint32_t) * (size_t)n->world));
    uint32_t *d_first = (uint32_t *)n->d_seam.p, *d_all = d_first + 1;
    uint64_t *d_off = (uint64_t *)(((uintptr_t)(d_all + n->world) + 7) & ~(uintptr_t)7);
    uint32_t *h_first = (uint32_t *)n->h_row.p; uint64_t *h_off = (uint64_t *)n->h_row.p + 1; uint32_t *h_all = (uint32_t *)(h_off + n->world);
    *h_first = pre != VLX_OK ? 0x100u : 0u;
    HIPCHK(c, hipMemcpyAsync(d_first, h_first, sizeof(uint32_t), hipMemcpyHostToDevice, st));
    if (pre == VLX_OK && shard_len) HIPCHK(c, hipMemcpyAsync(d_first, d_shard, 1, hipMemcpyDeviceToDevice, st));
    NCHK(n, n->coll->all_gather(d_first, d_all, sizeof(uint32_t), st, c->err));
    HIPCHK(c, hipMemcpyAsync(h_all, d_all, sizeof(uint32_t) * (size_t)n->world, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    if (pre != VLX_OK) return pre;
    for (int r = 0; r < n->world; r++) if (h_all[r] & 0x100u) { c->err = "another rank refused the gather"; return VLX_ERR_PEER; }
    if (is_root) HIPCHK(c, hipMemsetAsync(d_full, 0, 1, st));             // the very first byte is only ever OR-ed into
    // the placement: inside one group; the group is CLOSED on every path, the first error is kept
    int rc = n->coll->group_start(c->err);
    if (rc == VLX_OK) {
        std::string first_err;
        if (is_root) {
            for (int r = 0; r < n->world; r++) {
                const uint64_t o = n->off[(size_t)r], ln = n->seg_len[(size_t)r];
                if (ln <= 1) continue;
                int e;
                if (r == root) {
                    const hipError_t he = hipMemcpyAsync(d_full + o + 1, d_shard + 1, ln - 1, hipMemcpyDeviceToDevice, st);
                    e = he == hipSuccess ? VLX_OK : VLX_ERR_HIP;
                    if (e != VLX_OK) c->err = std::string("hipMemcpyAsync(own shard): ") + hipGetErrorString(he);
                } else e = n->coll->recv(d_full + o + 1, ln - 1, r, st, c->err);
                if (e != VLX_OK && rc == VLX_OK) { rc = e; first_err = c->err; }
            }
        } else if (shard_len > 1) {
            rc = n->coll->send(d_shard + 1, shard_len - 1, root, st, c->err);
            if (rc != VLX_OK) first_err = c->err;
        }
        const int e = n->coll->group_end(st, c->err);
        if (rc != VLX_OK) c->err = first_err; else rc = e;
    }
    if (rc == VLX_OK && is_root) {
        memcpy(h_off, n->off.data(), sizeof(uint64_t) * (size_t)n->world);
        HIPCHK(c, hipMemcpyAsync(d_off, h_off, sizeof(uint64_t) * (size_t)n->world, hipMemcpyHostToDevice, st));
        HIPCHK(c, vlx_launch_or_seams(st, d_full, d_off, d_all, (uint32_t)n->world));
    }
    HIPCHK(c, hipStreamSynchronize(st));
    return rc;
}

int vlx_node_all_gather_lengths(vlx_node *n, const uint64_t *h_lengths, uint32_t n_local, uint32_t max_per_rank,
                                uint64_t *h_table, uint32_t *h_counts)
{
    if (!n) return VLX_ERR_ARG;
    vlx_ctx *c = n->ctx;
    // (max_per_rank is the row length: every rank passes the same value; a rank with bad arguments sends a row that says so)
    const bool bad = (n_local && !h_lengths) || !h_table || !h_counts || n_local > max_per_rank;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t row = 1 + (size_t)max_per_rank;                          // count, lengths
    HIPCHK(c, n->h_row.ensure(sizeof(uint64_t) * row)); HIPCHK(c, n->h_table.ensure(sizeof(uint64_t) * row * (size_t)n->world));
    HIPCHK(c, n->d_row.ensure(sizeof(uint64_t) * row)); HIPCHK(c, n->d_table.ensure(sizeof(uint64_t) * row * (size_t)n->world));
    uint64_t *hr = (uint64_t *)n->h_row.p;
    memset(hr, 0, sizeof(uint64_t) * row);
    hr[0] = bad ? ~0ull : n_local;
    if (!bad) for (uint32_t i = 0; i < n_local; i++) hr[1 + i] = h_lengths[i];
    hipStream_t st = c->stream;
    HIPCHK(c, hipMemcpyAsync(n->d_row.p, hr, sizeof(uint64_t) * row, hipMemcpyHostToDevice, st));
    NCHK(n, n->coll->all_gather(n->d_row.p, n->d_table.p, sizeof(uint64_t) * row, st, c->err));
    HIPCHK(c, hipMemcpyAsync(n->h_table.p, n->d_table.p, sizeof(uint64_t) * row * (size_t)n->world, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    if (bad) { c->err = "bad arguments (n_local > max_per_rank or a null table)"; return VLX_ERR_ARG; }
    const uint64_t *tab = (const uint64_t *)n->h_table.p;
    for (int r = 0; r < n->world; r++) if (tab[(size_t)r * row] == ~0ull) { c->err = "another rank refused the length exchange"; return VLX_ERR_PEER; }
    for (int r = 0; r < n->world; r++) {
        h_counts[r] = (uint32_t)std::min<uint64_t>(tab[(size_t)r * row], max_per_rank);
        for (uint32_t i = 0; i < max_per_rank; i++) h_table[(size_t)r * max_per_rank + i] = i < h_counts[r] ? tab[(size_t)r * row + 1 + i] : 0;
    }
    return VLX_OK;
}

// For transports written where no HIP binding is at hand (velox_amd/sharded.py carries the collectives over any
// torch.distributed backend this way): wait for `stream`, then copy n bytes between any two host / device addresses.
int vlx_node_copy(void *dst, const void *src, size_t n, void *stream)
{
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return VLX_ERR_HIP;
    return !n || hipMemcpy(dst, src, n, hipMemcpyDefault) == hipSuccess ? VLX_OK : VLX_ERR_HIP;
}

// ---- decode partitioning: the chunk list of one host walk, cut into contiguous runs ------------------
int vlx_node_decode_range(const vlx_chunk *chunks, size_t n_chunks, int rank, int world, vlx_decode_range *out)
{
    if (!out || (!chunks && n_chunks) || world < 1 || rank < 0 || rank >= world) return VLX_ERR_ARG;
    const size_t base = n_chunks / (size_t)world, extra = n_chunks % (size_t)world;
    const size_t c0 = (size_t)rank * base + std::min<size_t>((size_t)rank, extra), c1 = c0 + base + ((size_t)rank < extra ? 1 : 0);
    memset(out, 0, sizeof(*out));
    out->chunk_lo = c0; out->chunk_hi = c1;
    if (c1 > c0) {
        // from the byte of the run's first u32 size prefix (a prefix cut short by the end of the data sits less than 32 bits
        // in front of its chunk: starting up to 4 bytes early is harmless, a negative start is not)
        const uint64_t b0 = chunks[c0].bit_off;
        out->byte_lo = b0 >= 32 ? (b0 - 32) >> 3 : 0;
        out->byte_hi = (chunks[c1 - 1].bit_off + chunks[c1 - 1].n_bits + 7) >> 3;
        out->out_lo = chunks[c0].out_off;
        out->out_n = chunks[c1 - 1].out_off + chunks[c1 - 1].n_out - out->out_lo;
    }
    return VLX_OK;
}

int vlx_node_decode_shard(vlx_ctx *c, const uint8_t *payload, size_t size, const vlx_chunk *chunks, size_t n_chunks, int rank, int world,
                          void *d_out, int out_fmt, vlx_decode_range *range)
{
    if (!c) return VLX_ERR_ARG;
    vlx_decode_range R;
    { const int rc = vlx_node_decode_range(chunks, n_chunks, rank, world, &R); if (rc != VLX_OK) return rc; }
    if (range) *range = R;
    if (R.chunk_hi == R.chunk_lo) return VLX_OK;
    if (!payload || !d_out || R.byte_hi > size) { c->err = "chunk run beyond the payload"; return VLX_ERR_ARG; }
    HIPCHK(c, hipSetDevice(c->device));
    const size_t nb = (size_t)(R.byte_hi - R.byte_lo);
    HIPCHK(c, c->dec_in.ensure(nb + 32));
    HIPCHK(c, hipMemcpyAsync(c->dec_in.p, payload + R.byte_lo, nb, hipMemcpyHostToDevice, c->stream));
    std::vector<vlx_chunk> mine(chunks + R.chunk_lo, chunks + R.chunk_hi);
    for (vlx_chunk &ch : mine) { ch.bit_off -= 8 * R.byte_lo; ch.out_off -= R.out_lo; }
    return vlx_decode_device(c, (const uint8_t *)c->dec_in.p, nb, mine.data(), mine.size(), d_out, out_fmt);
}

} // extern "C"

// ================================================================================================
// The pool: ProcessBlock / the DecodeNext loop over several GPUs from ONE caller thread (the reference's only
// caller is one process, main.cpp:136-137 / :253-261).  One worker thread + one vlx_ctx + one local-group node per
// entry of the device list; a device may be listed more than once (several shards in flight on one GPU).
struct vlx_pool {
    int n = 0;
    std::vector<vlx_ctx *> ctx; std::vector<vlx_node *> node; vlx_local_group *grp = nullptr;
    std::vector<std::thread> th;
    std::mutex m; std::condition_variable cv_go, cv_done;
    std::function<int(int)> job; uint64_t job_gen = 0; int pending = 0; bool stop = false;
    std::vector<int> rc; std::string err;
    // shared between the workers of one call
    uint8_t *out = nullptr; std::vector<uint8_t> firsts; std::vector<uint64_t> offs;

    void worker(int r)
    {
        uint64_t seen = 0;
        for (;;) {
            std::function<int(int)> j;
            { std::unique_lock<std::mutex> lk(m); cv_go.wait(lk, [&] { return stop || job_gen != seen; }); if (stop) return; seen = job_gen; j = job; }
            const int res = j(r);
            { std::lock_guard<std::mutex> lk(m); rc[(size_t)r] = res; if (--pending == 0) cv_done.notify_all(); }
        }
    }
    // runs `f(rank)` on every worker and waits; returns the first error (lowest rank)
    int run(std::function<int(int)> f)
    {
        std::unique_lock<std::mutex> lk(m);
        job = std::move(f); pending = n; job_gen++;
        cv_go.notify_all();
        cv_done.wait(lk, [&] { return pending == 0; });
        for (int r = 0; r < n; r++) if (rc[(size_t)r] != VLX_OK) { err = std::string("rank ") + std::to_string(r) + ": " + vlx_last_error(ctx[(size_t)r]); return rc[(size_t)r]; }
        return VLX_OK;
    }
};

extern "C" {

void vlx_pool_destroy(vlx_pool *p)
{
    if (!p) return;
    { std::lock_guard<std::mutex> lk(p->m); p->stop = true; }
    p->cv_go.notify_all();
    for (auto &t : p->th) if (t.joinable()) t.join();
    for (vlx_node *nd : p->node) vlx_node_destroy(nd);
    for (vlx_ctx *c : p->ctx) vlx_ctx_destroy(c);
    vlx_local_group_destroy(p->grp);
    delete p;
}

int vlx_pool_create(const int *devices, int n, vlx_pool **out)
{
    if (!out || n < 1 || !devices) return VLX_ERR_ARG;
    *out = nullptr;
    vlx_pool *p = new (std::nothrow) vlx_pool();
    if (!p) return VLX_ERR_NOMEM;
    p->n = n; p->rc.assign((size_t)n, VLX_OK); p->firsts.assign((size_t)n, 0); p->offs.assign((size_t)n, 0);
    int rc = vlx_local_group_create(n, &p->grp);
    for (int r = 0; r < n && rc == VLX_OK; r++) {
        vlx_ctx *c = nullptr;
        rc = vlx_ctx_create(devices[r], &c);
        if (rc != VLX_OK) break;
        p->ctx.push_back(c);
        vlx_node *nd = nullptr;
        rc = vlx_node_init_local(c, p->grp, r, &nd);
        if (rc == VLX_OK) p->node.push_back(nd);
    }
    if (rc != VLX_OK) { vlx_pool_destroy(p); return rc; }
    for (int r = 0; r < n; r++) p->th.emplace_back([p, r] { p->worker(r); });
    *out = p;
    return VLX_OK;
}

int vlx_pool_size(const vlx_pool *p) { return p ? p->n : 0; }
const char *vlx_pool_last_error(const vlx_pool *p) { return p ? p->err.c_str() : "null pool"; }

// Integer PCM of any container width (fmt = 16 / 24 / 32 / VLX_FMT_I64) in host memory, even total, at least one
// sub-block per shard worth cutting: rank r uploads its run of sub-blocks, the node layer does the two exchanges,
// rank r downloads its bytes straight into their place in the malloc'd payload.
static int pool_encode_int(vlx_pool *p, const uint8_t *in, size_t n_samples, int fmt, uint8_t **payload, size_t *payload_len)
{
    const size_t W = fmt == VLX_FMT_I64 ? 8 : (size_t)fmt / 8;
    const uint32_t bound = (uint32_t)((vlx_track_chunks(n_samples) + (uint64_t)p->n - 1) / (uint64_t)p->n);
    p->out = nullptr;
    std::atomic<uint64_t> total{0};
    const int rc = p->run([&](int r) -> int {
        vlx_ctx *c = p->ctx[(size_t)r]; vlx_node *nd = p->node[(size_t)r];
        uint64_t s0 = 0, ns = 0;
        vlx_node_shard_range(n_samples, r, p->n, &s0, &ns);
        int pre = VLX_OK;
        const size_t cap = vlx_encode_bound(&ns, 1) + 256;
        if (hipSetDevice(c->device) != hipSuccess || c->in_stage.ensure(ns * W + 64) != hipSuccess || c->out_stage.ensure(cap) != hipSuccess) { c->err = "staging allocation failed"; pre = VLX_ERR_NOMEM; }
        if (pre == VLX_OK && ns && hipMemcpyAsync(c->in_stage.p, in + s0 * W, ns * W, hipMemcpyHostToDevice, c->stream) != hipSuccess) { c->err = "upload of the shard failed"; pre = VLX_ERR_HIP; }
        uint64_t ln = 0, off = 0, tot = 0; int hr = 0;
        // (a rank that could not stage still makes the collective calls: null buffers are refused collectively)
        int e = vlx_node_encode_track(nd, pre == VLX_OK ? c->in_stage.p : nullptr, fmt, ns ? ns : 0, bound, pre == VLX_OK ? (uint8_t *)c->out_stage.p : nullptr,
                                      c->out_stage.cap, pre == VLX_OK ? &ln : nullptr, &off, &tot, &hr);
        if (pre != VLX_OK) e = pre;
        // rank 0 allocates the payload once every rank knows its length; everybody meets twice around that
        if (r == 0 && e == VLX_OK) { p->out = (uint8_t *)calloc(tot ? tot : 1, 1); total = tot; }
        if (!p->grp->barrier()) return e != VLX_OK ? e : VLX_ERR_PEER;
        if (e != VLX_OK) return e;
        if (!p->out) { c->err = "payload allocation failed"; return VLX_ERR_NOMEM; }
        p->offs[(size_t)r] = off; p->firsts[(size_t)r] = 0;
        if (ln) {
            if (hipMemcpy(&p->firsts[(size_t)r], c->out_stage.p, 1, hipMemcpyDeviceToHost) != hipSuccess) { c->err = "download of the shard failed"; return VLX_ERR_HIP; }
            if (ln > 1 && hipMemcpy(p->out + off + 1, (const uint8_t *)c->out_stage.p + 1, ln - 1, hipMemcpyDeviceToHost) != hipSuccess) { c->err = "download of the shard failed"; return VLX_ERR_HIP; }
        }
        return VLX_OK;
    });
    if (rc != VLX_OK) { free(p->out); p->out = nullptr; return rc; }
    for (int r = 0; r < p->n; r++) p->out[p->offs[(size_t)r]] |= p->firsts[(size_t)r];   // the byte two neighbouring shards share
    *payload = p->out; *payload_len = (size_t)total.load();
    p->out = nullptr;
    return VLX_OK;
}

// one shard is worth a GPU of its own from a few sub-blocks on; below that (and for what does not shard: odd
// totals, float tracks, whose prologue looks at the whole input) worker 0's context does the whole call
static bool pool_worth_sharding(const vlx_pool *p, size_t n_samples) { return p->n > 1 && !(n_samples & 1) && vlx_track_chunks(n_samples) >= (uint64_t)p->n; }

int vlx_pool_encode_pcm(vlx_pool *p, const uint8_t *pcm, size_t n_samples, int bits, uint8_t **payload, size_t *payload_len)
{
    if (!p || !payload || !payload_len || (n_samples && !pcm)) return VLX_ERR_ARG;
    if (bits != 16 && bits != 24 && bits != 32) return VLX_ERR_ARG;
    *payload = nullptr; *payload_len = 0;
    if (pool_worth_sharding(p, n_samples)) return pool_encode_int(p, pcm, n_samples, bits, payload, payload_len);
    return p->run([&](int r) { return r ? VLX_OK : vlx_encode_pcm(p->ctx[0], pcm, n_samples, bits, payload, payload_len); });
}

int vlx_pool_encode_block(vlx_pool *p, int64_t *samples, size_t n_samples, int is_float, const uint8_t *exps, size_t n_exps,
                          const uint8_t *raw_bytes, uint8_t **payload, size_t *payload_len)
{
    if (!p || !payload || !payload_len || (n_samples && !samples)) return VLX_ERR_ARG;
    *payload = nullptr; *payload_len = 0;
    if (!is_float && pool_worth_sharding(p, n_samples)) return pool_encode_int(p, (const uint8_t *)samples, n_samples, VLX_FMT_I64, payload, payload_len);
    return p->run([&](int r) { return r ? VLX_OK : vlx_encode_block(p->ctx[0], samples, n_samples, is_float, exps, n_exps, raw_bytes, payload, payload_len); });
}

int vlx_pool_decode_pcm(vlx_pool *p, const uint8_t *payload, size_t size, size_t total, int bits, uint8_t *pcm, size_t cap, size_t *n_decoded)
{
    if (n_decoded) *n_decoded = 0;
    if (!p || (!payload && size) || (!pcm && total)) return VLX_ERR_ARG;
    if (bits != 16 && bits != 24 && bits != 32) return VLX_ERR_ARG;
    const size_t W = (size_t)bits / 8;
    if (total > cap / W) { p->err = "pcm buffer too small"; return VLX_ERR_CAPACITY; }
    if (!total) return VLX_OK;
    const size_t max_chunks = total / VLX_SUB_BLOCK + 3;
    std::vector<vlx_chunk> chs(max_chunks);
    const long n = vlx_walk_chunks(payload, size, total, 0, 0, chs.data(), chs.size());   // one host walk (VeloxCore.h:338-353)
    if (n < 0) return (int)n;
    if (p->n == 1 || (size_t)n < (size_t)p->n)
        return p->run([&](int r) { return r ? VLX_OK : vlx_decode_pcm(p->ctx[0], payload, size, total, bits, pcm, cap, n_decoded); });
    memset(pcm, 0, total * W);                                            // samples the stream does not reach (main.cpp:254-261)
    std::atomic<size_t> done{0};
    const int rc = p->run([&](int r) -> int {
        vlx_ctx *c = p->ctx[(size_t)r];
        vlx_decode_range R;
        int e = vlx_node_decode_range(chs.data(), (size_t)n, r, p->n, &R);
        if (e != VLX_OK || R.chunk_hi == R.chunk_lo) return e;
        HIPCHK(c, hipSetDevice(c->device));
        HIPCHK(c, c->dec_out.ensure((size_t)R.out_n * W + 32));
        HIPCHK(c, hipMemsetAsync(c->dec_out.p, 0, (size_t)R.out_n * W, c->stream));
        e = vlx_node_decode_shard(c, payload, size, chs.data(), (size_t)n, r, p->n, c->dec_out.p, bits, nullptr);
        if (e != VLX_OK) return e;
        HIPCHK(c, hipMemcpy(pcm + (size_t)R.out_lo * W, c->dec_out.p, (size_t)R.out_n * W, hipMemcpyDeviceToHost));
        done += (size_t)R.out_n;
        return VLX_OK;
    });
    if (rc == VLX_OK && n_decoded) *n_decoded = done.load();
    return rc;
}

} // extern "C"
