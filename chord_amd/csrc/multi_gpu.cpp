// Multi-GPU entry points of the C ABI: the frame sharded by 64x64 screen tiles (tile_layout.cpp), the
// exchanges of DESIGN.md 6 (the owned tiles' HZB texels mid-frame; their HZB texels and visibility words at the end) issued
// by the library itself, so that a C++ host calls ONE function per frame like DeferredRenderer::render does
// (renderer.cpp:319-345).  The reference is single-device (graphics.cpp:524-548); nothing here has a counterpart in it.
//
// Two forms:
//   ChordGroup            one process, n devices, one host thread per device.  The exchange is a DIRECT all-gather: every
//                         rank pushes its chunk to each peer with its own hipMemcpyPeerAsync on a (source, destination)
//                         stream -- n-1 concurrent copies per rank, one per xGMI link of the fully connected node; a
//                         ring would serialise n-1 hops on one link.  Device ordinals may repeat, which is
//                         how the protocol (events, ordering, buffer reuse) is exercised on a one-GPU box.
//   chordvis_comm_*       one process per GPU (torch.distributed / MPI hosts): an RCCL communicator attached to a sharded
//                         context; chordvis_render_frame then runs phase a -> ncclAllGather -> phase b -> ncclAllGather
//                         -> phase c on the context's stream.  librccl is resolved at run time (dlopen), preferring the
//                         copy already loaded in the process: a PyTorch-ROCm wheel bundles its own RCCL next to its own
//                         HIP runtime, and a second HIP runtime in one process cannot open the device.

#include "device_layer.h"

#include <dlfcn.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

using namespace chord;

// ------------------------------------------------------------------------------------------ RCCL, resolved at run time
namespace {

struct NcclUniqueId { char internal[CHORDVIS_UNIQUE_ID_BYTES]; };
typedef struct ncclComm* NcclComm;
enum { kNcclSuccess = 0 };
enum { kNcclUint8 = 1, kNcclUint64 = 5 };                       // ncclDataType_t (rccl.h): ncclUint8 = 1, ncclUint64 = 5

struct Rccl {
    void* handle = nullptr;
    std::string origin;
    int (*GetVersion)(int*) = nullptr;
    int (*GetUniqueId)(NcclUniqueId*) = nullptr;
    int (*CommInitRank)(NcclComm*, int, NcclUniqueId, int) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, NcclComm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};

std::mutex gRcclMutex;
Rccl gRccl;
std::string gRcclError;

const Rccl* rccl()
{
    std::lock_guard<std::mutex> lk(gRcclMutex);
    if (gRccl.handle) return &gRccl;
    // a copy already mapped into the process wins (same HIP runtime as the host's); then the loader's search path
    const char* names[] = {"librccl.so", "librccl.so.1"};
    void* h = nullptr;
    std::string origin;
    if (const char* forced = getenv("CHORDVIS_RCCL")) { h = dlopen(forced, RTLD_NOW | RTLD_GLOBAL); origin = forced; }
    for (int pass = 0; pass < 2 && !h; pass++)
        for (const char* n : names) {
            h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
            if (h) { origin = std::string(n) + (pass == 0 ? " (already loaded)" : ""); break; }
        }
    if (!h) {
        const char* e = dlerror();                                // (one call: dlerror() clears the message it returns)
        gRcclError = std::string("librccl.so not found: ") + (e ? e : "");
        return nullptr;
    }
    Rccl r;
    r.handle = h; r.origin = origin;
#define SYM(field, name) r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, name))
    SYM(GetVersion, "ncclGetVersion"); SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy"); SYM(AllGather, "ncclAllGather"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    if (!r.GetVersion || !r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather) {
        gRcclError = "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather";
        return nullptr;
    }
    gRccl = r;
    return &gRccl;
}

int nccl_fail(ChordCtx* c, const Rccl* r, const char* what, int code)
{
    char buf[256];
    std::snprintf(buf, sizeof(buf), "%s: %s (%d)", what, r && r->GetErrorString ? r->GetErrorString(code) : "RCCL error", code);
    return fail(c, CHORDVIS_E_COMM, buf);
}

} // namespace

// ------------------------------------------------------------------------------------- one process per GPU: RCCL comm
extern "C" {

int chordvis_comm_unique_id(void* out128)
{
    if (!out128) return CHORDVIS_E_INVALID;
    const Rccl* r = rccl();
    if (!r) return CHORDVIS_E_COMM;
    NcclUniqueId id;
    if (r->GetUniqueId(&id) != kNcclSuccess) return CHORDVIS_E_COMM;
    std::memcpy(out128, &id, sizeof(id));
    return CHORDVIS_OK;
}

int chordvis_comm_init_rank(ChordCtx* c, uint32_t nranks, uint32_t rank, const void* id128)
{
    if (!c || !id128 || nranks == 0 || rank >= nranks) return fail(c, CHORDVIS_E_INVALID, "comm_init_rank: bad arguments");
    if (c->shard.ranks != nranks || c->shard.rank != rank)
        return fail(c, CHORDVIS_E_INVALID, "comm_init_rank: call chordvis_set_shard(nranks, rank) first (same nranks / rank)");
    const Rccl* r = rccl();
    if (!r) return fail(c, CHORDVIS_E_COMM, gRcclError.c_str());
    if (c->comm) { (void)r->CommDestroy((NcclComm)c->comm); c->comm = nullptr; }
    CHORD_HIP(c, hipSetDevice(c->device));
    NcclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    NcclComm comm = nullptr;
    const int rc = r->CommInitRank(&comm, (int)nranks, id, (int)rank);
    if (rc != kNcclSuccess) return nccl_fail(c, r, "ncclCommInitRank", rc);
    c->comm = comm;
    return CHORDVIS_OK;
}

static void comm_drop_pipeline(ChordCtx* c, const Rccl* r)
{
    if (c->commResolveStream) (void)hipStreamSynchronize(c->commResolveStream);
    if (c->commBulk && r) (void)r->CommDestroy((NcclComm)c->commBulk);
    c->commBulk = nullptr;
    if (c->commResolveStream) { (void)hipStreamDestroy(c->commResolveStream); c->commResolveStream = nullptr; }
    if (c->commPhaseB) { (void)hipEventDestroy(c->commPhaseB); c->commPhaseB = nullptr; }
    for (int k = 0; k < 2; k++) if (c->commVisReady[k]) {
        if (c->visReadyEvent[0] == c->commVisReady[k]) c->visReadyEvent[0] = nullptr;
        if (c->visReadyEvent[1] == c->commVisReady[k]) c->visReadyEvent[1] = nullptr;
        (void)hipEventDestroy(c->commVisReady[k]); c->commVisReady[k] = nullptr;
    }
    c->commPipelined = false;
}

int chordvis_comm_destroy(ChordCtx* c)
{
    if (!c) return CHORDVIS_E_INVALID;
    if (c->comm || c->commBulk) {
        const Rccl* r = rccl();
        (void)hipStreamSynchronize(c->stream);
        comm_drop_pipeline(c, r);
        if (r && c->comm) (void)r->CommDestroy((NcclComm)c->comm);
        c->comm = nullptr;
    }
    return CHORDVIS_OK;
}

// Pipelined frames over RCCL (the ChordGroup form: chordvis_group_set_pipelined).  id128: a SECOND unique id, for the
// communicator that carries the 66 MB image of frame i on a stream of its own while frame i + 1 is rendered; NULL switches the
// protocol off again (after the frames in flight have drained).
int chordvis_comm_set_pipelined(ChordCtx* c, const void* id128)
{
    if (!c || !c->comm) return fail(c, CHORDVIS_E_INVALID, "comm_set_pipelined: chordvis_comm_init_rank must come first");
    const Rccl* r = rccl();
    if (!r) return fail(c, CHORDVIS_E_COMM, gRcclError.c_str());
    CHORD_HIP(c, hipSetDevice(c->device));
    CHORD_HIP(c, hipStreamSynchronize(c->stream));
    comm_drop_pipeline(c, r);
    if (!id128) return CHORDVIS_OK;
    if (c->visExternal) return fail(c, CHORDVIS_E_INVALID, "comm_set_pipelined: the context must own its visibility buffer (two frames are alive at once)");
    NcclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    NcclComm comm = nullptr;
    const int rc = r->CommInitRank(&comm, (int)c->shard.ranks, id, (int)c->shard.rank);
    if (rc != kNcclSuccess) return nccl_fail(c, r, "ncclCommInitRank (image communicator)", rc);
    c->commBulk = comm;
    CHORD_HIP(c, hipStreamCreateWithFlags(&c->commResolveStream, hipStreamNonBlocking));
    CHORD_HIP(c, hipEventCreateWithFlags(&c->commPhaseB, hipEventDisableTiming));
    for (int k = 0; k < 2; k++) CHORD_HIP(c, hipEventCreateWithFlags(&c->commVisReady[k], hipEventDisableTiming));
    c->commPipelined = true;
    c->commFrameSerial = 0;
    return CHORDVIS_OK;
}

int chordvis_comm_info(ChordCtx* c, int* ncclVersion, uint32_t* nranks, char* libraryOrigin, uint32_t originBytes)
{
    const Rccl* r = rccl();
    if (!r) return c ? fail(c, CHORDVIS_E_COMM, gRcclError.c_str()) : CHORDVIS_E_COMM;
    if (ncclVersion) { int v = 0; (void)r->GetVersion(&v); *ncclVersion = v; }
    if (nranks) *nranks = (c && c->comm) ? c->shard.ranks : 0u;
    if (libraryOrigin && originBytes) std::snprintf(libraryOrigin, originBytes, "%s", r->origin.c_str());
    return CHORDVIS_OK;
}

} // extern "C"

namespace chord {

// The sharded frame of a context with a communicator: everything on the context's stream, no host synchronisation.
// A rank whose phase fails (a capacity or HIP error on that rank only) still issues BOTH collectives: its peers are already
// inside them on their streams and would wait forever otherwise.  Whether the mid-frame exchange happens is decided before
// phase a from state every rank shares (frame history + flags), like the ChordGroup path; the first error is returned after
// the last collective.
// ---- pipelined frame of one rank, shared by the two transports ---------------------------------------------------------------
// (chordvis_comm_set_pipelined over RCCL, chordvis_group_set_pipelined over peer copies)  The image of frame i is
// gathered on the side -- behind the rank's "phase b done" point, on the resolve stream -- and copied to row-major there, while
// the compute stream goes on to the history HZB, which needs only the small end-of-frame exchange (the tiles' HZB texels, valid
// ranges and bin lengths), and then to frame i + 1, which rasters into the other buffer pair.  The pair a frame takes over was
// last used two frames ago; the caller has waited for that frame's "image complete" event (and, in a group, for every rank to
// have done so) before calling.  ONE body, so that the sequence the one-device group tests exercise with 2..8 ranks is the
// sequence an RCCL host runs; a transport supplies
//   small(which, chunkBytes)         all-gather of an exchange buffer of the context (0: mid-frame, 1: end of frame), ordered on the
//                                    compute stream
//   image()                          all-gather of the current visibility buffer behind everything enqueued so far on the compute
//                                    stream, such that `resolveStream` may read the complete buffer afterwards
// A failing step is remembered; the rank still walks through every exchange (its peers are inside them).
template <class Transport>
int pipelined_frame_body(ChordCtx* c, hipEvent_t visReadyThis, hipEvent_t visReadyOther, hipStream_t resolveStream, Transport& tr)
{
    int rc = CHORDVIS_OK;
    { const int e = chordvis_swap_visibility(c); if (!rc) rc = e; }
    c->visReadyEvent[0] = visReadyThis;
    c->visReadyEvent[1] = visReadyOther;
    const bool stage1 = c->historySlot != 0 && (c->hView.flags & CHORD_FLAG_HZB_CULL);
    // the sharded group cull (2..8 ranks, flat mode: state every rank shares): each rank tests its share of the groups, the rank
    // masks are all-gathered, phase a goes on from them
    if (cull_shardable(c)) {
        if (!rc) rc = chordvis_frame_phase_cull(c);
        const int e = tr.small(2, (size_t)c->cullChunkBlocks * 257u * 4u); if (!rc) rc = e;
    }
    if (!rc) rc = chordvis_frame_phase_a(c);
    if (stage1) { const int e = tr.small(0, (size_t)c->hzbExchangeChunkHalves * 2); if (!rc) rc = e; }
    if (!rc) rc = chordvis_frame_phase_b(c);
    {
        int e = tr.image();
        if (!rc) rc = e;
        if (!rc) rc = chordvis_frame_resolve_visibility(c, resolveStream);
        const hipError_t he = hipEventRecord(visReadyThis, resolveStream);
        if (he != hipSuccess && !rc) rc = fail(c, CHORDVIS_E_HIP, "hipEventRecord(image complete)", he);
    }
    { const int e = tr.small(1, (size_t)c->hzbFinalExchangeChunkBytes); if (!rc) rc = e; }
    if (!rc) rc = chordvis_frame_phase_c_finish(c);
    return rc;
}

// the rank-major exchange buffers of a context by number: 0 = mid-frame (min chain after stage 0), 1 = end of frame, 2 = the
// sharded cull's rank masks
static char* exchange_buffer(ChordCtx* c, int which)
{
    return which == 0 ? reinterpret_cast<char*>(c->dHzbExchange) : which == 1 ? reinterpret_cast<char*>(c->dHzbFinalExchange) : reinterpret_cast<char*>(c->dCullExchange);
}

struct RcclTransport {
    ChordCtx* c; const Rccl* r;
    int small(int which, size_t bytes)
    {
        char* base = exchange_buffer(c, which);
        const int e = r->AllGather(base + (size_t)c->shard.rank * bytes, base, bytes, kNcclUint8, (NcclComm)c->comm, c->stream);
        return e == kNcclSuccess ? CHORDVIS_OK : nccl_fail(c, r, "ncclAllGather(exchange buffer)", e);
    }
    int image()
    {
        hipError_t he = hipEventRecord(c->commPhaseB, c->stream);
        if (he == hipSuccess) he = hipStreamWaitEvent(c->commResolveStream, c->commPhaseB, 0);
        if (he != hipSuccess) return fail(c, CHORDVIS_E_HIP, "image gather: stream order", he);
        const size_t words = c->shard.ranks > 1 ? (size_t)c->shard.slotsPerRank * (CHORD_TILE * CHORD_TILE) : (size_t)c->visWords;
        const int e = r->AllGather(c->dVis + (size_t)c->shard.rank * words, c->dVis, words, kNcclUint64, (NcclComm)c->commBulk, c->commResolveStream);
        return e == kNcclSuccess ? CHORDVIS_OK : nccl_fail(c, r, "ncclAllGather(visibility)", e);
    }
};

static int comm_render_frame_pipelined(ChordCtx* c, const Rccl* r)
{
    const int parity = (int)(c->commFrameSerial++ & 1u);
    const hipError_t he = hipEventSynchronize(c->commVisReady[parity]);      // (a never-recorded event reads as complete)
    RcclTransport tr{c, r};
    const int rc = pipelined_frame_body(c, c->commVisReady[parity], c->commVisReady[parity ^ 1], c->commResolveStream, tr);
    return he != hipSuccess ? fail(c, CHORDVIS_E_HIP, "hipEventSynchronize(image of two frames ago)", he) : rc;
}

int comm_render_frame(ChordCtx* c)
{
    const Rccl* r = rccl();
    if (!r || !c->comm) return fail(c, CHORDVIS_E_COMM, "render_frame: sharded context without a communicator (chordvis_comm_init_rank, or drive chordvis_frame_phase_a/b/c)");
    if (c->shard.ranks == 1) {
        // a communicator of one rank: the single-GPU frame (nothing is sharded), then the image through the collective in place --
        // what a one-GPU box can exercise of this path
        void* comm = c->comm;
        hipError_t he = hipSuccess;
        const int parity = (int)(c->commFrameSerial & 1u);
        if (c->commPipelined) {
            // (one buffer, not a pair: the frame may not start before the image gather of the frame before has read it)
            c->commFrameSerial++;
            he = hipStreamWaitEvent(c->stream, c->commVisReady[parity ^ 1], 0);
        }
        c->comm = nullptr;                                                    // (chordvis_render_frame dispatches on it)
        int rc = chordvis_render_frame(c);
        c->comm = comm;
        if (he != hipSuccess && !rc) rc = fail(c, CHORDVIS_E_HIP, "pipelined frame, one rank: stream order", he);
        if (c->commPipelined) {
            // the pipelined protocol's image step with the transport a host of N ranks uses: second communicator, resolve
            // stream behind the compute stream's "phase b done" event, "image complete" event per frame parity
            RcclTransport tr{c, r};
            const int e = tr.image();
            if (!rc) rc = e;
            he = hipEventRecord(c->commVisReady[parity], c->commResolveStream);
            if (he != hipSuccess && !rc) rc = fail(c, CHORDVIS_E_HIP, "hipEventRecord(image complete)", he);
            c->visReadyEvent[0] = c->commVisReady[parity];
            c->visReadyEvent[1] = c->commVisReady[parity ^ 1];
            return rc;
        }
        const int e = r->AllGather(c->dVis, c->dVis, (size_t)c->visWords, kNcclUint64, (NcclComm)comm, c->stream);
        if (e != kNcclSuccess && !rc) rc = nccl_fail(c, r, "ncclAllGather(visibility, one rank)", e);
        return rc;
    }
    if (c->commPipelined) return comm_render_frame_pipelined(c, r);
    const bool stage1 = c->historySlot != 0 && (c->hView.flags & CHORD_FLAG_HZB_CULL);
    int rc = CHORDVIS_OK;
    if (cull_shardable(c)) {
        // the sharded group cull (decided from state every rank shares -- configuration and a buffer made at set-up, never from an
        // allocation inside the frame): this rank's share of the group tests, then the rank masks of all ranks
        rc = chordvis_frame_phase_cull(c);
        const size_t bytes = (size_t)c->cullChunkBlocks * 257u * 4u;
        char* base = reinterpret_cast<char*>(c->dCullExchange);
        const int e = r->AllGather(base + (size_t)c->shard.rank * bytes, base, bytes, kNcclUint8, (NcclComm)c->comm, c->stream);
        if (e != kNcclSuccess && !rc) rc = nccl_fail(c, r, "ncclAllGather(cull exchange)", e);
    }
    if (!rc) rc = chordvis_frame_phase_a(c);
    if (stage1) {
        // RCCL has no 16-bit integer type; the payload is opaque f16 bits
        const size_t bytes = (size_t)c->hzbExchangeChunkHalves * 2;
        char* base = reinterpret_cast<char*>(c->dHzbExchange);
        const int e = r->AllGather(base + (size_t)c->shard.rank * bytes, base, bytes, kNcclUint8, (NcclComm)c->comm, c->stream);
        if (e != kNcclSuccess && !rc) rc = nccl_fail(c, r, "ncclAllGather(mid-frame HZB exchange)", e);
    }
    if (!rc) rc = chordvis_frame_phase_b(c);
    {
        const size_t bytes = (size_t)c->hzbFinalExchangeChunkBytes;
        char* base = reinterpret_cast<char*>(c->dHzbFinalExchange);
        int e = r->AllGather(base + (size_t)c->shard.rank * bytes, base, bytes, kNcclUint8, (NcclComm)c->comm, c->stream);
        if (e != kNcclSuccess && !rc) rc = nccl_fail(c, r, "ncclAllGather(end-of-frame HZB exchange)", e);
        stamp(c, S_EXCH_FINAL);                             // (the segment up to phase c's stamp is then the image gather alone)
        const size_t words = (size_t)c->shard.slotsPerRank * (CHORD_TILE * CHORD_TILE);
        e = r->AllGather(c->dVis + (size_t)c->shard.rank * words, c->dVis, words, kNcclUint64, (NcclComm)c->comm, c->stream);
        if (e != kNcclSuccess && !rc) rc = nccl_fail(c, r, "ncclAllGather(visibility)", e);
    }
    if (!rc) rc = chordvis_frame_phase_c(c);
    return rc;
}

} // namespace chord

// ------------------------------------------------------------------------------------------ one process, n devices
struct ChordGroup {
    uint32_t n = 0;
    std::vector<ChordCtx*> ctx;
    std::vector<int> device;
    std::string lastError;
    // copy streams and events: index [src * n + dst]
    std::vector<hipStream_t> copyStream;
    std::vector<hipEvent_t> evArrived[4];       // per exchange (0 = HZB exchanges, 1 = visibility; 2 / 3 = visibility of a pipelined frame, by parity)
    std::vector<hipEvent_t> evReady[4];         // per rank
    // pipelined mode: the visibility all-gather + row-major copy of frame i run beside frame i + 1 (per-rank resolve stream,
    // "image complete" event per buffer parity)
    bool pipelined = false;
    uint64_t frameSerial = 0;
    std::vector<hipStream_t> resolveStream;
    std::vector<hipStream_t> bulkCopyStream;    // [src * n + dst]: the travelling image has its own copy streams -- the small HZB exchanges of the same and the next frame must not queue behind it
    std::vector<hipEvent_t> evVisReady[2];
    // worker threads: one per rank, parked on a condition variable between jobs
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cvJob, cvDone;
    std::function<int(uint32_t)> job;
    uint64_t jobSerial = 0;
    uint32_t pending = 0;
    std::vector<int> jobRc;
    bool quit = false;
    // host time each worker spent inside its chordvis_group_render_frame job (enqueueing kernels, copies, waits for its peers'
    // generations), accumulated until chordvis_group_enqueue_ms reads it: what the call costs the host, rank by rank
    std::vector<double> frameJobMs;
    std::vector<uint32_t> frameJobs;
    // An event must be RECORDED before another thread enqueues a wait on it.  Round 2 put two host barriers (mutex + condition
    // variable over all rank threads) into every exchange; now a rank PUBLISHES what it has recorded as a generation number per
    // event and a peer spins (no sleep: the threads run concurrently and are microseconds apart) until the generation it needs
    // is there -- pairwise, lock-free, no wake-up latency.  Generations count the uses of an event channel, in lockstep on every
    // rank; an event is re-recorded for generation k + 1 only after every consumer of generation k has enqueued its wait
    // (group_all_gather, "reuse").
    std::unique_ptr<std::atomic<uint64_t>[]> readyGen[4];     // [which][rank]
    std::unique_ptr<std::atomic<uint64_t>[]> arrivedGen[4];   // [which][src * n + dst]
    std::vector<uint64_t> useCount[4];                         // [which][rank]: private to the rank's thread
    std::atomic<bool> abort{false};                            // a hand-shake timed out: every exchange of the group fails from now on
    // one barrier remains, once per pipelined frame (the swap of the buffer pairs)
    std::mutex bm;
    std::condition_variable bcv;
    uint32_t bCount = 0; uint64_t bGen = 0;
};

namespace {

void group_barrier(ChordGroup* g)
{
    std::unique_lock<std::mutex> lk(g->bm);
    const uint64_t gen = g->bGen;
    if (++g->bCount == g->n) { g->bCount = 0; g->bGen++; g->bcv.notify_all(); }
    else g->bcv.wait(lk, [&] { return g->bGen != gen; });
}

void worker_main(ChordGroup* g, uint32_t rank)
{
    (void)hipSetDevice(g->device[rank]);
    uint64_t seen = 0;
    for (;;) {
        std::function<int(uint32_t)> job;
        {
            std::unique_lock<std::mutex> lk(g->m);
            g->cvJob.wait(lk, [&] { return g->quit || g->jobSerial != seen; });
            if (g->quit) return;
            seen = g->jobSerial;
            job = g->job;
        }
        const int rc = job(rank);
        {
            std::lock_guard<std::mutex> lk(g->m);
            g->jobRc[rank] = rc;
            if (--g->pending == 0) g->cvDone.notify_all();
        }
    }
}

// runs fn(rank) on every rank's thread; returns the first failure (and keeps its message)
int run_all(ChordGroup* g, const std::function<int(uint32_t)>& fn, const char* what)
{
    {
        std::unique_lock<std::mutex> lk(g->m);
        g->job = fn; g->pending = g->n; g->jobSerial++;
        g->cvJob.notify_all();
        g->cvDone.wait(lk, [&] { return g->pending == 0; });
    }
    for (uint32_t r = 0; r < g->n; r++)
        if (g->jobRc[r]) {
            char buf[640];
            std::snprintf(buf, sizeof(buf), "%s: rank %u (device %d): %s", what, r, g->device[r], chordvis_last_error(g->ctx[r]));
            g->lastError = buf;
            return g->jobRc[r];
        }
    return CHORDVIS_OK;
}

// Bounded: a peer that never publishes (its thread died, or it skipped an exchange because a decision every rank must share
// diverged) must not leave this thread spinning forever.  The first rank to give up raises the group-wide abort flag; every
// spin of every rank then ends at once, the frame returns CHORDVIS_E_COMM on all of them and the group stays poisoned until it
// is destroyed (its generation counters no longer agree).
// The limit is long on purpose (CHORDVIS_GROUP_TIMEOUT_S, default 300 s): a slow peer is legitimate -- a first frame under a profiler,
// code-object load, a multi-second frame, N ranks time-sharing one device -- and an abort poisons the group.
inline double group_timeout_seconds()
{
    static const double s = [] { const char* e = getenv("CHORDVIS_GROUP_TIMEOUT_S"); const double v = e ? atof(e) : 0.0; return v > 0.0 ? v : 300.0; }();
    return s;
}

inline bool spin_until(const std::atomic<uint64_t>& a, uint64_t gen, std::atomic<bool>& abort)
{
    uint32_t spins = 0;
    std::chrono::steady_clock::time_point t0;
    while (a.load(std::memory_order_acquire) < gen) {
        if (abort.load(std::memory_order_relaxed)) return false;
        if (++spins > 4096u) {
            if (spins == 4097u) t0 = std::chrono::steady_clock::now();
            std::this_thread::yield();
            if ((spins & 0xFFFu) == 0u && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > group_timeout_seconds()) { abort.store(true); return false; }
        }
    }
    return true;
}

// Direct all-gather of rank-major buffers, called by every rank's thread: rank r's chunk travels to each peer on the
// (r, d) copy stream once both ends are ready -- r has produced it, d has finished with the region it lands in (d's
// stream is past every earlier reader of its buffer when it records `ready`).  A compute stream then waits for every
// copy that ends in its buffer AND every copy that leaves it (the source region is rewritten by the next frame).
// Host side: no barrier.  Reuse of the channel's events: ready[r] of use k + 1 is recorded after r saw arrived[o -> r] of use
// k from every peer o (o records it after enqueueing its waits on ready[r]); arrived[r -> d] of use k + 1 is recorded after r
// saw ready[d] of use k + 1, which d publishes after it enqueued, in use k, its waits on arrived[r -> d].
int group_all_gather(ChordGroup* g, uint32_t r, int which, const std::function<char*(uint32_t)>& base, size_t chunkBytes, hipStream_t waiter = nullptr,
                     bool bulk = false)
{
    ChordCtx* c = g->ctx[r];
    if (!waiter) waiter = c->stream;                              // who continues once the buffer is complete (default: the rank's compute stream)
    const uint32_t n = g->n;
    int rc = CHORDVIS_OK;
    // a failing call is remembered, but the rank keeps publishing its generations: its peers spin on them
#define GG_HIP(call) do { const hipError_t e_ = (call); if (e_ != hipSuccess && !rc) rc = fail(c, CHORDVIS_E_HIP, #call, e_); } while (0)
    const uint64_t gen = ++g->useCount[which][r];
    GG_HIP(hipEventRecord(g->evReady[which][r], c->stream));
    g->readyGen[which][r].store(gen, std::memory_order_release);
    bool alive = true;
    for (uint32_t k = 1; k < n && alive; k++) {
        const uint32_t d = (r + k) % n;                           // (every rank starts with a different peer)
        if (!spin_until(g->readyGen[which][d], gen, g->abort)) { alive = false; break; }
        hipStream_t cs = bulk ? g->bulkCopyStream[(size_t)r * n + d] : g->copyStream[(size_t)r * n + d];
        GG_HIP(hipStreamWaitEvent(cs, g->evReady[which][r], 0));
        GG_HIP(hipStreamWaitEvent(cs, g->evReady[which][d], 0));
        GG_HIP(hipMemcpyPeerAsync(base(d) + (size_t)r * chunkBytes, g->device[d], base(r) + (size_t)r * chunkBytes, g->device[r], chunkBytes, cs));
        GG_HIP(hipEventRecord(g->evArrived[which][(size_t)r * n + d], cs));
        g->arrivedGen[which][(size_t)r * n + d].store(gen, std::memory_order_release);
    }
    for (uint32_t k = 1; k < n && alive; k++) {
        const uint32_t o = (r + k) % n;
        if (!spin_until(g->arrivedGen[which][(size_t)o * n + r], gen, g->abort)) { alive = false; break; }
        GG_HIP(hipStreamWaitEvent(waiter, g->evArrived[which][(size_t)o * n + r], 0));   // into my buffer
        GG_HIP(hipStreamWaitEvent(waiter, g->evArrived[which][(size_t)r * n + o], 0));   // out of my buffer
    }
#undef GG_HIP
    if (!alive && !rc) rc = fail(c, CHORDVIS_E_COMM, "group exchange: a peer never reached the hand-shake (timed out or aborted); destroy the group");
    return rc;
}

int gfail(ChordGroup* g, int code, const char* what) { if (g) g->lastError = what; return code; }

struct JobClock {      // a worker's time inside one frame job (its own slot: no lock)
    ChordGroup* g; uint32_t r; std::chrono::steady_clock::time_point t0;
    JobClock(ChordGroup* g_, uint32_t r_) : g(g_), r(r_), t0(std::chrono::steady_clock::now()) {}
    ~JobClock() { g->frameJobMs[r] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); g->frameJobs[r]++; }
};

} // namespace

extern "C" {

int chordvis_create_group(uint32_t n, const int* deviceOrdinals, ChordGroup** out)
{
    if (!out) return CHORDVIS_E_INVALID;
    *out = nullptr;
    if (n == 0 || n > 64 || !deviceOrdinals) return CHORDVIS_E_INVALID;
    ChordGroup* g = new ChordGroup();
    g->n = n;
    g->device.assign(deviceOrdinals, deviceOrdinals + n);
    g->ctx.assign(n, nullptr);
    g->jobRc.assign(n, 0);
    g->frameJobMs.assign(n, 0.0); g->frameJobs.assign(n, 0);
    int rc = CHORDVIS_OK;
    for (uint32_t r = 0; r < n && !rc; r++) rc = chordvis_create(g->device[r], nullptr, &g->ctx[r]);
    g->copyStream.assign((size_t)n * n, nullptr);
    g->bulkCopyStream.assign((size_t)n * n, nullptr);
    for (int w = 0; w < 4; w++) {
        g->evArrived[w].assign((size_t)n * n, nullptr); g->evReady[w].assign(n, nullptr);
        g->readyGen[w].reset(new std::atomic<uint64_t>[n]); g->arrivedGen[w].reset(new std::atomic<uint64_t>[(size_t)n * n]);
        for (uint32_t i = 0; i < n; i++) g->readyGen[w][i].store(0);
        for (size_t i = 0; i < (size_t)n * n; i++) g->arrivedGen[w][i].store(0);
        g->useCount[w].assign(n, 0);
    }
    g->resolveStream.assign(n, nullptr);
    for (int w = 0; w < 2; w++) g->evVisReady[w].assign(n, nullptr);
    for (uint32_t r = 0; r < n && !rc; r++) {
        if (hipSetDevice(g->device[r]) != hipSuccess) { rc = CHORDVIS_E_NO_DEVICE; break; }
        for (uint32_t d = 0; d < n; d++) {
            if (d != r && g->device[d] != g->device[r]) {
                int can = 0;
                (void)hipDeviceCanAccessPeer(&can, g->device[r], g->device[d]);
                if (can) { const hipError_t e = hipDeviceEnablePeerAccess(g->device[d], 0); if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) rc = CHORDVIS_E_HIP; (void)hipGetLastError(); }
            }
            if (d != r && hipStreamCreateWithFlags(&g->copyStream[(size_t)r * n + d], hipStreamNonBlocking) != hipSuccess) rc = CHORDVIS_E_HIP;
            if (d != r && hipStreamCreateWithFlags(&g->bulkCopyStream[(size_t)r * n + d], hipStreamNonBlocking) != hipSuccess) rc = CHORDVIS_E_HIP;
            for (int w = 0; w < 4; w++)
                if (d != r && hipEventCreateWithFlags(&g->evArrived[w][(size_t)r * n + d], hipEventDisableTiming) != hipSuccess) rc = CHORDVIS_E_HIP;
        }
        for (int w = 0; w < 4; w++) if (hipEventCreateWithFlags(&g->evReady[w][r], hipEventDisableTiming) != hipSuccess) rc = CHORDVIS_E_HIP;
        if (hipStreamCreateWithFlags(&g->resolveStream[r], hipStreamNonBlocking) != hipSuccess) rc = CHORDVIS_E_HIP;
        for (int w = 0; w < 2; w++) if (hipEventCreateWithFlags(&g->evVisReady[w][r], hipEventDisableTiming) != hipSuccess) rc = CHORDVIS_E_HIP;
    }
    if (rc) { chordvis_destroy_group(g); return rc; }
    for (uint32_t r = 0; r < n; r++) g->workers.emplace_back(worker_main, g, r);
    *out = g;
    return CHORDVIS_OK;
}

int chordvis_destroy_group(ChordGroup* g)
{
    if (!g) return CHORDVIS_E_INVALID;
    {
        std::lock_guard<std::mutex> lk(g->m);
        g->quit = true;
        g->cvJob.notify_all();
    }
    for (std::thread& t : g->workers) t.join();
    for (uint32_t r = 0; r < g->n; r++) if (g->ctx[r]) (void)hipStreamSynchronize(g->ctx[r]->stream);
    for (hipStream_t s : g->copyStream) if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
    for (hipStream_t s : g->bulkCopyStream) if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
    for (hipStream_t s : g->resolveStream) if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
    for (int w = 0; w < 4; w++) {
        for (hipEvent_t e : g->evArrived[w]) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : g->evReady[w]) if (e) (void)hipEventDestroy(e);
    }
    for (ChordCtx* c : g->ctx) if (c) { c->visReadyEvent[0] = c->visReadyEvent[1] = nullptr; }   // (the group's events)
    for (int w = 0; w < 2; w++) for (hipEvent_t e : g->evVisReady[w]) if (e) (void)hipEventDestroy(e);
    for (ChordCtx* c : g->ctx) if (c) chordvis_destroy(c);
    delete g;
    return CHORDVIS_OK;
}

uint32_t chordvis_group_size(ChordGroup* g) { return g ? g->n : 0u; }
ChordCtx* chordvis_group_ctx(ChordGroup* g, uint32_t rank) { return (g && rank < g->n) ? g->ctx[rank] : nullptr; }
const char* chordvis_group_last_error(ChordGroup* g) { return g ? g->lastError.c_str() : "null group"; }

int chordvis_group_set_limits(ChordGroup* g, const ChordLimits* limits)
{
    if (!g || !limits) return gfail(g, CHORDVIS_E_INVALID, "group_set_limits: null argument");
    return run_all(g, [&](uint32_t r) { return chordvis_set_limits(g->ctx[r], limits); }, "group_set_limits");
}

int chordvis_group_upload_scene(ChordGroup* g, const ChordSceneDesc* scene)
{
    if (!g || !scene) return gfail(g, CHORDVIS_E_INVALID, "group_upload_scene: null argument");
    return run_all(g, [&](uint32_t r) { return chordvis_upload_scene(g->ctx[r], scene); }, "group_upload_scene");   // replicated
}

int chordvis_group_allocate_gbuffer(ChordGroup* g, uint32_t width, uint32_t height)
{
    if (!g) return CHORDVIS_E_INVALID;
    return run_all(g, [&](uint32_t r) {
        int rc = chordvis_set_shard(g->ctx[r], g->n, r);
        if (!rc) rc = chordvis_allocate_gbuffer(g->ctx[r], width, height, nullptr);
        return rc;
    }, "group_allocate_gbuffer");
}

int chordvis_group_update_objects(ChordGroup* g, const ChordObject* hostObjects, uint32_t count)
{
    if (!g || !hostObjects) return gfail(g, CHORDVIS_E_INVALID, "group_update_objects: null argument");
    return run_all(g, [&](uint32_t r) { return chordvis_update_objects(g->ctx[r], hostObjects, count); }, "group_update_objects");
}

int chordvis_group_set_view(ChordGroup* g, const ChordCameraView* view, const ChordInstanceCullingView* iv, uint32_t switchFlags)
{
    if (!g || !view || !iv) return gfail(g, CHORDVIS_E_INVALID, "group_set_view: null argument");
    for (uint32_t r = 0; r < g->n; r++) {                      // host-only state: no thread hop needed
        const int rc = chordvis_set_view(g->ctx[r], view, iv, switchFlags);
        if (rc) { g->lastError = chordvis_last_error(g->ctx[r]); return rc; }
    }
    return CHORDVIS_OK;
}

static int group_render_frame_pipelined(ChordGroup* g);

int chordvis_group_render_frame(ChordGroup* g)
{
    if (!g) return CHORDVIS_E_INVALID;
    if (g->n == 1) return run_all(g, [&](uint32_t r) { return chordvis_render_frame(g->ctx[r]); }, "group_render_frame");
    if (g->pipelined) return group_render_frame_pipelined(g);
    // A rank that fails keeps walking through the host barriers (its peers would wait for it forever otherwise).
    // (whether the frame starts with the sharded cull is ONE decision for the group, taken here before any worker runs)
    bool shardCull = true;
    for (uint32_t k = 0; k < g->n; k++) shardCull = shardCull && cull_shardable(g->ctx[k]);
    return run_all(g, [&](uint32_t r) {
        ChordCtx* c = g->ctx[r];
        JobClock clock(g, r);
        // the same on every rank (they share the frame history), and taken BEFORE phase a so that a rank whose phase a
        // fails still joins the exchange its peers are about to enter
        const bool stage1 = c->historySlot != 0 && (c->hView.flags & CHORD_FLAG_HZB_CULL);
        int rc = CHORDVIS_OK;
        if (shardCull) {
            rc = chordvis_frame_phase_cull(c);
            const int e = group_all_gather(g, r, 0, [&](uint32_t k) { return reinterpret_cast<char*>(g->ctx[k]->dCullExchange); },
                                           (size_t)c->cullChunkBlocks * 257u * 4u);
            if (!rc) rc = e;
        }
        if (!rc) rc = chordvis_frame_phase_a(c);
        if (stage1) {
            const int e = group_all_gather(g, r, 0, [&](uint32_t k) { return reinterpret_cast<char*>(g->ctx[k]->dHzbExchange); },
                                           (size_t)c->hzbExchangeChunkHalves * 2);
            if (!rc) rc = e;
        }
        if (!rc) rc = chordvis_frame_phase_b(c);
        {
            int e = group_all_gather(g, r, 0, [&](uint32_t k) { return reinterpret_cast<char*>(g->ctx[k]->dHzbFinalExchange); },
                                     (size_t)c->hzbFinalExchangeChunkBytes);
            if (!rc) rc = e;
            chord::stamp(c, chord::S_EXCH_FINAL);
            e = group_all_gather(g, r, 1, [&](uint32_t k) { return reinterpret_cast<char*>(g->ctx[k]->dVis); }, (size_t)c->shard.slotsPerRank * (CHORD_TILE * CHORD_TILE) * 8);
            if (!rc) rc = e;
        }
        if (!rc) rc = chordvis_frame_phase_c(c);
        return rc;
    }, "group_render_frame");
}

// Pipelined frames (VERDICT r01 item 2 / DESIGN.md 6): the visibility all-gather of frame i -- 58 of 66 MB arriving per rank
// at 4K -- and its row-major copy leave the frame's critical path and run beside frame i + 1:
//   * two buffer pairs per rank (chordvis_swap_visibility): frame i + 1 rasters into the other one;
//   * the history HZB of frame i needs only the small end-of-frame exchange (the tiles' HZB texels out of the tile kernel),
//     never the gathered image: chordvis_frame_phase_c_finish;
//   * the gather itself waits on the rank's "phase b done" event, travels on the (source, destination) copy streams, and is
//     followed by the row-major copy on the rank's resolve stream; "image complete" is an event per buffer pair that
//     the read-back / consumer entry points of the context wait for.
// A buffer pair is reused two frames later; by then its gather has long finished -- the host waits for it before the swap.
int chordvis_group_set_pipelined(ChordGroup* g, int enable)
{
    if (!g) return CHORDVIS_E_INVALID;
    if (g->n < 2 && enable) return gfail(g, CHORDVIS_E_INVALID, "group_set_pipelined: needs at least two ranks");
    const int rc = chordvis_group_sync(g);
    if (rc) return rc;
    g->pipelined = enable != 0;
    return CHORDVIS_OK;
}

struct GroupTransport {
    ChordGroup* g; uint32_t r; int parity;
    int small(int which, size_t bytes)
    {
        return group_all_gather(g, r, 0, [&, which](uint32_t k) { return chord::exchange_buffer(g->ctx[k], which); }, bytes);
    }
    int image()
    {
        ChordCtx* c = g->ctx[r];
        const int e = group_all_gather(g, r, 2 + parity, [&](uint32_t k) { return reinterpret_cast<char*>(g->ctx[k]->dVis); },
                                       (size_t)c->shard.slotsPerRank * (CHORD_TILE * CHORD_TILE) * 8, g->resolveStream[r], true);
        // (the resolve stream also needs the rank's own chunk: the `ready` event of this exchange was recorded on the compute
        // stream after phase b)
        const hipError_t he = hipStreamWaitEvent(g->resolveStream[r], g->evReady[2 + parity][r], 0);
        if (he != hipSuccess && !e) return fail(c, CHORDVIS_E_HIP, "image gather: own chunk", he);
        return e;
    }
};

static int group_render_frame_pipelined(ChordGroup* g)
{
    const uint64_t serial = g->frameSerial++;
    const int parity = (int)(serial & 1u);
    return run_all(g, [&, parity](uint32_t r) {
        ChordCtx* c = g->ctx[r];
        JobClock clock(g, r);
        // the buffer pair this frame takes over was last used two frames ago: its gather and copy are complete on every rank
        // once every rank has seen its "image complete" event of that frame (recorded behind the waits for all copies into
        // AND out of the rank's buffer).  NOT a drain of the resolve stream: the previous frame's image is still travelling,
        // and this frame's kernels are to be enqueued beside it.
        const hipError_t he = hipEventSynchronize(g->evVisReady[parity][r]);
        group_barrier(g);
        GroupTransport tr{g, r, parity};
        const int rc = chord::pipelined_frame_body(c, g->evVisReady[parity][r], g->evVisReady[parity ^ 1][r], g->resolveStream[r], tr);
        return he != hipSuccess ? fail(c, CHORDVIS_E_HIP, "hipEventSynchronize(image of two frames ago)", he) : rc;
    }, "group_render_frame (pipelined)");
}

// Re-balances the tile map of every rank from the last frame's loads (chordvis_rebalance on each context: the ranks hold the
// same loads and compute the same map).  Drains the frames in flight first.
int chordvis_group_rebalance(ChordGroup* g, uint32_t* imbalancePermille)
{
    if (!g) return CHORDVIS_E_INVALID;
    if (g->n < 2) { if (imbalancePermille) *imbalancePermille = 1000u; return CHORDVIS_OK; }
    int rc = chordvis_group_sync(g);
    if (rc) return rc;
    std::vector<uint32_t> imb(g->n, 1000u);
    rc = run_all(g, [&](uint32_t r) { return chordvis_rebalance(g->ctx[r], &imb[r]); }, "group_rebalance");
    if (imbalancePermille) *imbalancePermille = imb[0];
    return rc;
}

// Mean host time per frame each rank's worker thread spent inside chordvis_group_render_frame since the last call (milliseconds,
// n values; then reset): the launches and copies it enqueued plus its waits for the peers' hand-shakes.
int chordvis_group_enqueue_ms(ChordGroup* g, double* msPerRank, uint32_t n)
{
    if (!g || !msPerRank || n != g->n) return gfail(g, CHORDVIS_E_INVALID, "group_enqueue_ms: one value per rank");
    std::lock_guard<std::mutex> lk(g->m);                          // (no job is running: run_all returned)
    for (uint32_t r = 0; r < n; r++) { msPerRank[r] = g->frameJobs[r] ? g->frameJobMs[r] / g->frameJobs[r] : 0.0; g->frameJobMs[r] = 0.0; g->frameJobs[r] = 0; }
    return CHORDVIS_OK;
}

int chordvis_group_sync(ChordGroup* g)
{
    if (!g) return CHORDVIS_E_INVALID;
    return run_all(g, [&](uint32_t r) {
        int rc = chordvis_sync(g->ctx[r]);
        if (!rc && g->resolveStream[r] && hipStreamSynchronize(g->resolveStream[r]) != hipSuccess) rc = fail(g->ctx[r], CHORDVIS_E_HIP, "group_sync: resolve stream", hipGetLastError());
        return rc;
    }, "group_sync");
}

} // extern "C"
