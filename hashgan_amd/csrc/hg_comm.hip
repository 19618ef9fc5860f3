// libhashgan_amd.so -- collectives: RCCL over xGMI on the context's own stream, the library dlopen'ed on first use.
#include "hg_ctx.hpp"

#include <dlfcn.h>
#include <mutex>
#include <unistd.h>

namespace {
// ---- RCCL, loaded on first use.  The product links no collective library: a single-GPU process never pays for it. ----
struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;

std::mutex g_rccl_mu;                              // contexts of several threads may initialise communicators at once

int rccl_load() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.handle) return HG_OK;
    const char* cands[] = {getenv("HG_RCCL_LIBRARY"), "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    void* h = nullptr;
    std::string tried;
    for (const char* name : cands) {
        if (!name || !*name) continue;
        h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
        const char* why = dlerror();
        tried += std::string(name) + ": " + (why ? why : "?") + "; ";
    }
    if (!h) return fail(HG_ERR_STATE, "RCCL not found (%s)", tried.c_str());
    RcclApi a;
    a.handle = h;
#define HG_SYM(field, name)                                                             \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(h, name));                      \
    if (!a.field) { dlclose(h); return fail(HG_ERR_STATE, "RCCL: symbol %s missing", name); }
    HG_SYM(GetUniqueId, "ncclGetUniqueId")
    HG_SYM(CommInitRank, "ncclCommInitRank")
    HG_SYM(CommDestroy, "ncclCommDestroy")
    HG_SYM(AllGather, "ncclAllGather")
    HG_SYM(AllReduce, "ncclAllReduce")
    HG_SYM(Send, "ncclSend")
    HG_SYM(Recv, "ncclRecv")
    HG_SYM(GroupStart, "ncclGroupStart")
    HG_SYM(GroupEnd, "ncclGroupEnd")
    HG_SYM(GetErrorString, "ncclGetErrorString")
#undef HG_SYM
    g_rccl = a;
    return HG_OK;
}

#define HG_NCCL(expr)                                                                                  \
    do {                                                                                               \
        ncclResult_t r_ = (expr);                                                                      \
        if (r_ != ncclSuccess)                                                                         \
            return fail(HG_ERR_HIP, "%s: %s (%s:%d)", #expr, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)
}  // namespace

void comm_release(hg_ctx* c) {
    if (c->comm && g_rccl.CommDestroy) { (void)g_rccl.CommDestroy(c->comm); c->comm = nullptr; }
}

extern "C" {

// ---- collectives: RCCL over xGMI, on the context's own stream (no PyTorch anywhere) ----------------------------
int hg_comm_unique_id(uint8_t* id) {
    if (!id) return fail(HG_ERR_ARG, "hg_comm_unique_id: null pointer");
    HG_TRY(rccl_load());
    ncclUniqueId u;
    HG_NCCL(g_rccl.GetUniqueId(&u));
    static_assert(sizeof u == HG_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(id, &u, sizeof u);
    return HG_OK;
}

int hg_comm_init(hg_ctx* c, const uint8_t* id, int rank, int world) {
    if (!c || !id) return fail(HG_ERR_ARG, "hg_comm_init: null argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(HG_ERR_ARG, "hg_comm_init: rank %d of %d", rank, world);
    if (c->comm) return fail(HG_ERR_STATE, "hg_comm_init: the context already has a communicator (hg_comm_destroy first)");
    HG_TRY(c->use());
    HG_TRY(rccl_load());
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    // RCCL prints a version banner on stdout when a communicator comes up; stdout belongs to the caller (bench.py's
    // one JSON line): send whatever the library prints during the call to stderr instead
    fflush(stdout);
    const int saved = dup(1);
    if (saved >= 0) (void)dup2(2, 1);
    const ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, u, rank);
    fflush(stdout);
    if (saved >= 0) { (void)dup2(saved, 1); close(saved); }
    if (r != ncclSuccess) { c->comm = nullptr; return fail(HG_ERR_HIP, "ncclCommInitRank: %s", g_rccl.GetErrorString(r)); }
    c->comm_rank = rank;
    c->comm_world = world;
    return HG_OK;
}

int hg_comm_destroy(hg_ctx* c) {
    if (!c) return fail(HG_ERR_ARG, "hg_comm_destroy: null context");
    if (!c->comm) return HG_OK;
    HG_TRY(c->use());
    HG_TRY(c->sync());
    ncclComm_t k = c->comm;
    c->comm = nullptr;
    c->comm_rank = 0; c->comm_world = 1;
    HG_NCCL(g_rccl.CommDestroy(k));
    return HG_OK;
}

int hg_comm_info(hg_ctx* c, int* rank, int* world) {
    if (!c) return fail(HG_ERR_ARG, "hg_comm_info: null context");
    if (rank) *rank = c->comm ? c->comm_rank : 0;
    if (world) *world = c->comm ? c->comm_world : 0;      // 0: no communicator
    return HG_OK;
}

int hg_allgather(hg_ctx* c, int slot, const void* dev_src, int64_t nbytes, void** dev_gathered) {
    if (!c || !dev_src || !dev_gathered || nbytes < 1) return fail(HG_ERR_ARG, "hg_allgather: bad argument");
    if (slot < 0 || slot >= 4) return fail(HG_ERR_ARG, "hg_allgather: slot %d outside 0..3", slot);
    if (!c->comm) return fail(HG_ERR_STATE, "hg_allgather: no communicator (hg_comm_init)");
    HG_TRY(c->use());
    DevBuf& out = c->gathered[slot];
    HG_TRY(out.reserve((size_t)nbytes * c->comm_world));
    c->t_begin(KI_COMM);
    const ncclResult_t rg = g_rccl.AllGather(dev_src, out.p, (size_t)nbytes, ncclUint8, c->comm, c->stream);
    c->t_end();                                        // (before a failure is reported: the timing slots stay balanced)
    if (rg != ncclSuccess) return fail(HG_ERR_HIP, "hg_allgather: ncclAllGather: %s", g_rccl.GetErrorString(rg));
    *dev_gathered = out.p;
    return c->stage_end();
}

// The owner-routed exchanges of the sharded bet: block r of dev_src (nbytes each) goes to rank r, *dev_out = [world][nbytes],
// block r = what rank r sent here.  One grouped ncclSend / ncclRecv per peer: point-to-point over xGMI, every link busy at once,
// and a GPU receives only its own queries' share of the tables (an all-gather would deliver all of them to everybody).
int hg_alltoall(hg_ctx* c, int slot, const void* dev_src, int64_t nbytes_per_peer, void** dev_out) {
    if (!c || !dev_src || !dev_out || nbytes_per_peer < 1) return fail(HG_ERR_ARG, "hg_alltoall: bad argument");
    if (slot < 0 || slot >= 4) return fail(HG_ERR_ARG, "hg_alltoall: slot %d outside 0..3", slot);
    if (!c->comm) return fail(HG_ERR_STATE, "hg_alltoall: no communicator (hg_comm_init)");
    HG_TRY(c->use());
    DevBuf& out = c->gathered[slot];
    HG_TRY(out.reserve((size_t)nbytes_per_peer * c->comm_world));
    c->t_begin(KI_COMM);
    // Whatever fails inside the group, the group is closed and the timing slot ended before the first failure is reported:
    // a communicator left inside an open group (or an unbalanced t_begin) would make every later collective on this
    // stream undefined.
    ncclResult_t first = g_rccl.GroupStart();
    const char* where = "ncclGroupStart";
    if (first == ncclSuccess) {
        for (int r = 0; r < c->comm_world && first == ncclSuccess; ++r) {
            first = g_rccl.Send((const char*)dev_src + (size_t)r * nbytes_per_peer, (size_t)nbytes_per_peer, ncclUint8, r, c->comm, c->stream);
            where = "ncclSend";
            if (first != ncclSuccess) break;
            first = g_rccl.Recv(out.as<char>() + (size_t)r * nbytes_per_peer, (size_t)nbytes_per_peer, ncclUint8, r, c->comm, c->stream);
            where = "ncclRecv";
        }
        const ncclResult_t ended = g_rccl.GroupEnd();
        if (first == ncclSuccess && ended != ncclSuccess) { first = ended; where = "ncclGroupEnd"; }
    }
    c->t_end();
    if (first != ncclSuccess) return fail(HG_ERR_HIP, "hg_alltoall: %s: %s", where, g_rccl.GetErrorString(first));
    *dev_out = out.p;
    return c->stage_end();
}

// ---- one rank's whole step of the database-sharded bet in ONE call -------------------------------------------------------------
// sampled histogram -> [all-to-all by query owner] -> the owners guess -> [all-to-all back] -> select + rank the shard's own records ->
// [all-to-all by query owner] -> merge + AP of this rank's queries -> [all-gather of 16 bytes per query] -> one download.
// Everything is enqueued back to back on the context's stream (kernels and RCCL alike): no host round trip before the final download,
// no Python between the stages -- sharded.evaluate_shard's twelve stage calls cost a rank ~0.1 ms of a 1 ms step at G = 8.
// The exchange: the context's RCCL communicator; without one, `replica_world` >= 1 stands in for it with device-to-device copies of
// this rank's own blocks (every peer a replica of this rank: tools/replica_shard_timing.py times what ONE rank of a G-GPU run
// executes; with replica_world = 1 the result is the one-GPU result, which the tests check).
namespace {
int shard_alltoall(hg_ctx* c, bool replica, int world, int rank, int slot, const void* src, int64_t per, void** out) {
    if (!replica) return hg_alltoall(c, slot, src, per, out);
    DevBuf& o = c->gathered[slot];
    HG_TRY(o.reserve((size_t)per * world));
    c->t_begin(KI_COMM);
    for (int r = 0; r < world; ++r)        // what replica r would send here = what this rank sends to itself
        HG_HIP(hipMemcpyAsync(o.as<char>() + (size_t)r * per, (const char*)src + (size_t)rank * per, (size_t)per, hipMemcpyDeviceToDevice, c->stream));
    c->t_end();
    *out = o.p;
    return HG_OK;
}
int shard_allgather(hg_ctx* c, bool replica, int world, int slot, const void* src, int64_t nbytes, void** out) {
    if (!replica) return hg_allgather(c, slot, src, nbytes, out);
    DevBuf& o = c->gathered[slot];
    HG_TRY(o.reserve((size_t)nbytes * world));
    c->t_begin(KI_COMM);
    for (int r = 0; r < world; ++r)
        HG_HIP(hipMemcpyAsync(o.as<char>() + (size_t)r * nbytes, src, (size_t)nbytes, hipMemcpyDeviceToDevice, c->stream));
    c->t_end();
    *out = o.p;
    return HG_OK;
}
int shard_step_enqueue(hg_ctx* c, int64_t R, bool replica, int world, int rank, double* host_ap, int64_t* host_rel, int* bet_lost) {
    void *p = nullptr, *recv = nullptr;
    int64_t n = 0;
    HG_TRY(hg_sample_hist(c, R));
    HG_TRY(hg_pack_sample_by_owner(c, world, &p, &n));
    HG_TRY(shard_alltoall(c, replica, world, rank, 0, p, n, &recv));
    HG_TRY(hg_guess_owned(c, R, recv, world, rank, &p, &n));
    HG_TRY(shard_alltoall(c, replica, world, rank, 1, p, n, &recv));
    HG_TRY(hg_guess_finish(c, R, recv, world, rank));
    HG_TRY(hg_select_ranked(c));
    HG_TRY(hg_pack_ranked_by_owner(c, world, &p, &n));
    HG_TRY(shard_alltoall(c, replica, world, rank, 2, p, n, &recv));
    HG_TRY(hg_merge_ap_owned(c, recv, world, rank, &p, &n));
    HG_TRY(shard_allgather(c, replica, world, 3, p, n, &recv));
    const int64_t width = make_owners(c->geo.Q, world).width;
    return hg_unpack_parts(c, recv, world, width, host_ap, host_rel, bet_lost);      // the step's one synchronisation
}
}  // namespace

int hg_shard_step(hg_ctx* c, int64_t R, int replica_world, double* host_ap, int64_t* host_rel, int* bet_lost) {
    if (!c || !host_ap || !host_rel || !bet_lost) return fail(HG_ERR_ARG, "hg_shard_step: null argument");
    const bool replica = !c->comm;
    if (replica && replica_world < 1) return fail(HG_ERR_STATE, "hg_shard_step: no communicator (hg_comm_init), and no replica world given");
    const int world = replica ? replica_world : c->comm_world, rank = replica ? 0 : c->comm_rank;
    int eligible = 0;
    HG_TRY(hg_bet_eligible(c, R, world, &eligible));
    // (hg_merge_ranked's limits: lane r <-> shard r, the four queries of a block keep their [G][b + 1] record counts in LDS)
    if (!eligible || world > 64 || (size_t)4 * world * (c->b + 1) * 4 > 160u * 1024u) { *bet_lost = -1; return HG_OK; }   // nothing enqueued: the staged sequences apply
    const bool sync0 = c->stage_sync;
    c->stage_sync = false;                             // the stages only enqueue; hg_unpack_parts waits once
    const int rc = shard_step_enqueue(c, R, replica, world, rank, host_ap, host_rel, bet_lost);
    c->stage_sync = sync0;
    return rc;
}

// The north star's exchange: every shard's ranked (dist, idx) lists all-gathered and merged (exactly one shard owns a
// slot, the others hold HG_IDX_NONE / 0xFF there).  hg_get_topr then returns the global lists on every rank.
int hg_allgather_topr(hg_ctx* c) {
    HG_TRY(need(c, ST_SELECT, "hg_allgather_topr", "hg_select / hg_rank"));
    if (!c->lists_valid) return fail(HG_ERR_STATE, "hg_allgather_topr: ranked lists were not materialised by the last call");
    if (!c->comm) return fail(HG_ERR_STATE, "hg_allgather_topr: no communicator (hg_comm_init)");
    const i64 n = (i64)c->geo.Q * c->geo.R;
    const int G = c->comm_world;
    HG_TRY(c->gath_idx.reserve((size_t)n * 4 * G));       // own landing zones: hg_allgather's slots may hold live data
    HG_TRY(c->gath_dist.reserve((size_t)n * G));
    c->t_begin(KI_COMM);
    ncclResult_t rt = g_rccl.AllGather(c->out_idx.p, c->gath_idx.p, (size_t)n * 4, ncclUint8, c->comm, c->stream);
    if (rt == ncclSuccess) rt = g_rccl.AllGather(c->out_dist.p, c->gath_dist.p, (size_t)n, ncclUint8, c->comm, c->stream);
    c->t_end();
    if (rt != ncclSuccess) return fail(HG_ERR_HIP, "hg_allgather_topr: ncclAllGather: %s", g_rccl.GetErrorString(rt));
    HG_TRY(launch_min_topr(c, c->gath_idx.as<u32>(), c->gath_dist.as<u8>(), n, G));
    return c->stage_end();
}

// max over the ranks of one host double (step times of a benchmark), and a barrier: both one tiny all-reduce
int hg_allreduce_max_f64(hg_ctx* c, double* host_inout) {
    if (!c || !host_inout) return fail(HG_ERR_ARG, "hg_allreduce_max_f64: null argument");
    if (!c->comm) return fail(HG_ERR_STATE, "hg_allreduce_max_f64: no communicator (hg_comm_init)");
    HG_TRY(c->use());
    HG_TRY(c->comm_tmp.reserve(16));
    HG_HIP(hipMemcpyAsync(c->comm_tmp.p, host_inout, 8, hipMemcpyHostToDevice, c->stream));
    HG_NCCL(g_rccl.AllReduce(c->comm_tmp.p, c->comm_tmp.as<char>() + 8, 1, ncclFloat64, ncclMax, c->comm, c->stream));
    HG_HIP(hipMemcpyAsync(host_inout, c->comm_tmp.as<char>() + 8, 8, hipMemcpyDeviceToHost, c->stream));
    return c->sync();
}

int hg_barrier(hg_ctx* c) {
    double x = 0.0;
    return hg_allreduce_max_f64(c, &x);
}

}  // extern "C"
