// C ABI entry points + the host-side mirror of the reference's pass surface.
//
//   chordvis_instance_culling      <- instanceCulling                 instance_culling.cpp:83-161
//   chordvis_hzb_culling           <- detail::hzbCulling              instance_culling.cpp:286-351
//   chordvis_render_mesh           <- renderMesh / renderMeshRasterPipe  mesh_raster.cpp:76-254
//   chordvis_visibility_stage0/1   <- gltfVisibilityRenderingStage0/1 mesh_raster.cpp:269-329
//   chordvis_build_hzb             <- buildHZB                        hzb.cpp:38-227
//   chordvis_render_frame          <- DeferredRenderer::render hot segment  renderer.cpp:315-345,489
//
// Host code only records work on one HIP stream (the reference records Vulkan commands into one
// command buffer); counts never come back to the CPU inside a frame.

#include "device_layer.h"

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstring>

using namespace chord;

namespace chord {

int fail(ChordCtx* ctx, int code, const char* what, hipError_t e)
{
    if (ctx) {
        char buf[512];
        if (e != hipSuccess) std::snprintf(buf, sizeof(buf), "%s: %s (%d)", what, hipGetErrorString(e), (int)e);
        else std::snprintf(buf, sizeof(buf), "%s", what);
        ctx->lastError = buf;
    }
    return code;
}

} // namespace chord

namespace {

template <typename T>
int dalloc(ChordCtx* c, T** p, size_t count)
{
    if (*p) { (void)hipFree(*p); *p = nullptr; }
    if (count == 0) count = 1;
    CHORD_HIP(c, hipMalloc((void**)p, count * sizeof(T)));
    return CHORDVIS_OK;
}

template <typename T>
void dfree(T*& p) { if (p) { (void)hipFree((void*)p); p = nullptr; } }

void record(ChordCtx* c, int tag) { chord::stamp(c, tag); }

void begin_frame_stamps(ChordCtx* c)
{
    c->frameLaunchBase = c->launchCount;
    c->stampThisFrame = c->timers && (c->frameIndex % c->timerPeriod) == 0;
    c->frameIndex++;
    if (!c->stampThisFrame) return;
    if (c->timers == 1) { c->stampTags.clear(); c->framesStamped = 0; }
    c->framesStamped++;
    chord::stamp(c, S_FRAME_BEGIN);
}

int alloc_hzb(ChordCtx* c, HzbBuffers& h)
{
    ChordHZBDesc d;
    if (chordvis_hzb_desc(c->width, c->height, &d) != CHORDVIS_OK) return fail(c, CHORDVIS_E_INVALID, "render size unsupported for HZB");
    h.desc = d;
    h.valid = false;
    int rc;
    if ((rc = dalloc(c, &h.minTexels, d.totalTexels))) return rc;
    if ((rc = dalloc(c, &h.maxTexels, d.totalTexels))) return rc;
    if ((rc = dalloc(c, &h.validRange, 2))) return rc;
    CHORD_HIP(c, hipMemsetAsync(h.minTexels, 0, sizeof(uint16_t) * d.totalTexels, c->stream));
    CHORD_HIP(c, hipMemsetAsync(h.maxTexels, 0, sizeof(uint16_t) * d.totalTexels, c->stream));
    return CHORDVIS_OK;
}

int configure_targets(ChordCtx* c, uint64_t* external)
{
    // visibility words (sharded: rank-major tile slots, ranks * slotsPerRank * 64 * 64 words)
    const uint32_t N = c->shard.ranks;
    c->tilesX = (c->width + CHORD_TILE - 1) >> CHORD_TILE_SHIFT; c->tilesY = (c->height + CHORD_TILE - 1) >> CHORD_TILE_SHIFT;
    c->shard.tilesX = c->tilesX;
    // (allocated for the slot capacity; an all-gather moves ranks x shard.slotsPerRank slots, what the current map uses)
    c->slotCapacity = N > 1 ? chordvis_tile_slot_capacity(c->width, c->height, N) : 0u;
    c->visWords = N > 1 ? (uint64_t)N * c->slotCapacity * (CHORD_TILE * CHORD_TILE) : (uint64_t)c->width * c->height;
    int rc;
    if (external) {
        dfree(c->dVisOwned);
        c->dVis = external; c->visExternal = true;
    } else {
        if ((rc = dalloc(c, &c->dVisOwned, c->visWords))) return rc;
        c->dVis = c->dVisOwned; c->visExternal = false;
    }
    if (N > 1) { if ((rc = dalloc(c, &c->dVisResolved, (uint64_t)c->width * c->height))) return rc; }
    else dfree(c->dVisResolved);
    for (int i = 0; i < 3; i++) if ((rc = alloc_hzb(c, c->hzb[i]))) return rc;
    c->historySlot = 0;
    c->pendingTailSlot = 0;
    {   // per-tile triangle bins of the rasterizer: 64x64-pixel tiles, binCap entries each
        c->binCap = CHORD_BIN_CAP;              // 4K: 2 passes x 2040 tiles x 16384 x 4 B = 267 MB
        if ((rc = dalloc(c, &c->dTileBins, (size_t)2 * c->tilesX * c->tilesY * c->binCap))) return rc;
        c->binPoolChunks = c->limitPoolChunks;  // default: 2 passes x 32 Ki chunks x 1024 entries x 4 B = 256 MB
        if ((rc = dalloc(c, &c->dBinPool, (size_t)2 * c->binPoolChunks * CHORD_BIN_CHUNK))) return rc;
        const size_t tabWords = (size_t)2 * c->tilesX * c->tilesY * c->binMaxChunks;
        if ((rc = dalloc(c, &c->dBinChunkTab, tabWords))) return rc;
        CHORD_HIP(c, hipMemset(c->dBinChunkTab, 0, tabWords * sizeof(unsigned long long)));
        // work items of the tile kernel: every tile once, plus the slices of split tiles (bounded by the entries
        // a pass can hold)
        const size_t tilesN = (size_t)c->tilesX * c->tilesY;
        c->tileItemCap = (uint32_t)(tilesN * (CHORD_TILE_MAX_SLICES + 1u));
        if ((rc = dalloc(c, &c->dTileOrder, ((size_t)1 + c->tileItemCap) * 2))) return rc;   // uint2 per item
        if ((rc = dalloc(c, &c->dTileOrderKeep, ((size_t)1 + c->tileItemCap) * 4))) return rc;    // (two schedules each: this frame's and the one being made for the next)
        if ((rc = dalloc(c, &c->dTileOrderKeep1, ((size_t)1 + c->tileItemCap) * 4))) return rc;
        c->orderAge = c->orderAge1 = 0xFFFFFFFFu;
        if ((rc = dalloc(c, &c->dTileSlabs, tilesN * CHORD_TILE * CHORD_TILE))) return rc;
        CHORD_HIP(c, hipMemset(c->dTileSlabs, 0, tilesN * CHORD_TILE * CHORD_TILE * sizeof(unsigned long long)));
    }
    if ((rc = dalloc(c, &c->dTileRange, (size_t)2 * CHORD_MAX_TILES))) return rc;
    if (!c->hBinHint) {     // what the tile order kernel tells the host about a pass (the longest bin): 64 bytes of mapped host memory
        void* h = nullptr; void* dh = nullptr;
        CHORD_HIP(c, hipHostMalloc(&h, 64, hipHostMallocMapped));
        std::memset(h, 0, 64);
        CHORD_HIP(c, hipHostGetDevicePointer(&dh, h, 0));
        c->hBinHint = static_cast<volatile uint32_t*>(h); c->dBinHint = static_cast<uint32_t*>(dh);
    }
    if (c->dHotTiles) CHORD_HIP(c, hipMemsetAsync(c->dHotTiles, 0, sizeof(uint32_t) * 2 * (1 + CHORD_HOT_TILES), c->stream));   // (tile ids of another target size)
    c->hBinHint[0] = 0u; c->hBinHint[1] = 0u;
    c->hBinHint[2] = 0xFFFFFFFFu; c->hBinHint[3] = 0xFFFFFFFFu;      // clusters per pass of the last finished frame: none yet
    for (int k = 4; k < 8; k++) c->hBinHint[k] = 0u;                  // serials of the last pass found heavy [4 + pass] / light [6 + pass] (launch_raster: TILE_DIRECT): none
    {   // one {min, max} partial per block of the mip-0 kernel (64 x 4 texels per block)
        const uint32_t vw = (c->width + 1) / 2, vh = (c->height + 1) / 2;
        if ((rc = dalloc(c, &c->dRangePartials, (size_t)((vw + 63) / 64) * ((vh + 3) / 4) * 2))) return rc;
    }
    // sharded: the tile map and the two HZB exchange buffers (one slot per tile slot of the visibility buffer)
    if (N > 1) {
        const size_t tilesN = (size_t)c->tilesX * c->tilesY;
        CHORD_HIP(c, hipMemsetAsync(c->dVis, 0, c->visWords * 8, c->stream));   // (slots of edge tiles are partly padding: defined bytes on the wire)
        if ((rc = dalloc(c, &c->dShardTables, 64 + (tilesN + 1) / 2))) return rc;
        if ((rc = dalloc(c, &c->dTileLoads, tilesN))) return rc;
        CHORD_HIP(c, hipMemsetAsync(c->dTileLoads, 0, tilesN * 4, c->stream));
        if (c->tileOwners.size() != tilesN || !c->tileOwnersExplicit) {
            c->tileOwners.assign(tilesN, 0);
            c->tileOwnersExplicit = false;
            if (tile_layout(c->tilesX, c->tilesY, N, nullptr, 0u, c->tileOwners.data()) != CHORDVIS_OK) return fail(c, CHORDVIS_E_INVALID, "tile layout");
        }
        const size_t slots = (size_t)N * c->slotCapacity;
        if ((rc = dalloc(c, &c->dHzbExchange, slots * CHORD_HZB_SLOT_HALVES))) return rc;
        CHORD_HIP(c, hipMemsetAsync(c->dHzbExchange, 0, slots * CHORD_HZB_SLOT_HALVES * 2, c->stream));
        if ((rc = dalloc(c, &c->dHzbFinalExchange, slots * CHORD_HZB_FINAL_SLOT_HALVES))) return rc;
        CHORD_HIP(c, hipMemsetAsync(c->dHzbFinalExchange, 0, slots * CHORD_HZB_FINAL_SLOT_HALVES * 2, c->stream));
        if ((rc = install_tile_owners(c))) return rc;
        if ((rc = prepare_cull_exchange(c))) return rc;
    } else {
        dfree(c->dHzbExchange); dfree(c->dHzbFinalExchange); dfree(c->dShardTables); dfree(c->dTileLoads); dfree(c->dTileOwner); dfree(c->dCullExchange);
        c->shard.ownedRows = nullptr; c->shard.tileSlot = nullptr;
        c->hzbExchangeHalves = c->hzbExchangeChunkHalves = 0; c->hzbFinalExchangeChunkBytes = 0;
    }
    // (a second pair of visibility buffers is made by chordvis_swap_visibility when a pipelined host first asks for it)
    dfree(c->dVisAlt); dfree(c->dVisResolvedAlt);
    return CHORDVIS_OK;
}

HzbBuffers from_handle(const ChordHZB* h)
{
    HzbBuffers b;
    b.desc = h->desc; b.minTexels = h->minTexels; b.maxTexels = h->maxTexels; b.validRange = h->validRange;
    b.valid = h->minTexels != nullptr;
    return b;
}

CmdList from_handle(const ChordCountAndCmd& h)
{
    CmdList l; l.count = h.count; l.cmds = h.cmds; l.capacity = h.capacity; return l;
}

int ready(ChordCtx* c, const char* fn)
{
    if (!c) return CHORDVIS_E_INVALID;
    if (!c->sceneLoaded || !c->viewSet || !c->dVis) {
        char buf[160];
        std::snprintf(buf, sizeof(buf), "%s: upload_scene, allocate_gbuffer and set_view must come first", fn);
        return fail(c, CHORDVIS_E_INVALID, buf);
    }
    return CHORDVIS_OK;
}

// addClearGbufferPass inside a frame: the visibility words are not memset; the first raster pass of the
// frame starts every 64x64 tile from zero in LDS and writes every tile back, which is the clear.
// passes other than instance_culling that run before it in a frame still need the constants on the device
int flush_view(ChordCtx* c)
{
    if (c->viewDirty) {
        CHORD_HIP(c, hipMemcpyAsync(c->dView, &c->hView, sizeof(DView), hipMemcpyHostToDevice, c->stream));
        c->viewDirty = false;
    }
    if (c->zeroFrameStateInCull) {
        CHORD_HIP(c, hipMemsetAsync(c->dFrameState, 0, c->frameStateZeroBytes, c->stream));
        c->zeroFrameStateInCull = false;
    }
    return CHORDVIS_OK;
}

// The tail of the last fused frame's history HZB (mips 6.. + valid range) normally rides on the next frame's first
// kernel; anything else that is about to read that chain launches it now.
int flush_pending_tail(ChordCtx* c)
{
    if (c->pendingTailSlot) {
        launch_hzb_tail(c, c->hzb[c->pendingTailSlot], true, true);
        c->pendingTailSlot = 0;
        CHORD_HIP(c, hipGetLastError());
    }
    return CHORDVIS_OK;
}

int begin_frame_clear(ChordCtx* c)
{
    // counters, the four command-list counts and both passes' tile bin counts: zeroed by the frame's first
    // kernel (object_cull), not by a separate memset
    c->frameStateZeroBytes = offsetof(FrameState, tileCount) + sizeof(uint32_t) * CHORD_TILECOUNT_STRIDE * ((size_t)2 * c->tilesX * c->tilesY);
    c->zeroFrameStateInCull = true;
    c->rasterCalls = 0;
    c->pendingClear = true;
    c->inFrame = true;
    return CHORDVIS_OK;
}

int do_raster(ChordCtx* c, const CmdList& in)
{
    // renderMesh (mesh_raster.cpp:208-254): the four (alphaMode x twoSided) pipeline buckets and
    // their filter passes collapse into one launch; the kernels read bTwoSided and alphaMode per cluster (masked
    // triangles carry their texture coordinates and are alpha-tested per pixel; blended materials draw nothing).
    const hipError_t e = launch_raster(c, in, c->pendingClear);
    c->pendingClear = false;
    if (e != hipSuccess) return fail(c, CHORDVIS_E_HIP, "launch_raster", e);
    CHORD_HIP(c, hipGetLastError());
    return CHORDVIS_OK;
}

} // namespace

namespace chord {
void stamp(ChordCtx* c, int tag)
{
    if (!c->timers || !c->stampThisFrame) return;
    const size_t i = c->stampTags.size();
    if (i >= c->evPool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return;
        c->evPool.push_back(e);
    }
    (void)hipEventRecord(c->evPool[i], c->stream);
    c->stampTags.push_back(tag);
}
}

namespace chord {
// c->tileOwners (host, the same on every rank) -> the two device tables the kernels read: this rank's tiles as one bit mask per
// tile row, and every tile's slot in the rank-major buffers (owner * slotsPerRank + index among the owner's tiles, by tile index).
int install_tile_owners(ChordCtx* c)
{
    const uint32_t N = c->shard.ranks, tiles = c->tilesX * c->tilesY;
    if (N <= 1 || !c->dShardTables || c->tileOwners.size() != tiles) return fail(c, CHORDVIS_E_INVALID, "tile owners: no sharded G-buffer");
    std::vector<uint32_t> used(N, 0);
    for (uint32_t t = 0; t < tiles; t++) {
        const uint32_t o = c->tileOwners[t];
        if (o >= N || ++used[o] > c->slotCapacity) return fail(c, CHORDVIS_E_INVALID, "tile owners: an owner out of range, or a rank with more tiles than chordvis_tile_slot_capacity");
    }
    // a rank's chunk: as many slots as the largest rank owns (every all-gather moves ranks x S slots)
    const uint32_t S = *std::max_element(used.begin(), used.end());
    std::vector<unsigned long long> tab(64 + (tiles + 1) / 2, 0ull);
    uint32_t* slot = reinterpret_cast<uint32_t*>(tab.data() + 64);
    std::fill(used.begin(), used.end(), 0u);
    for (uint32_t t = 0; t < tiles; t++) {
        const uint32_t o = c->tileOwners[t];
        slot[t] = o * S + used[o]++;
        if (o == c->shard.rank) tab[t / c->tilesX] |= 1ull << (t % c->tilesX);
    }
    c->shard.slotsPerRank = S;
    c->hzbExchangeChunkHalves = (uint64_t)S * CHORD_HZB_SLOT_HALVES;
    c->hzbExchangeHalves = c->hzbExchangeChunkHalves * N;
    c->hzbFinalExchangeChunkBytes = (uint64_t)S * CHORD_HZB_FINAL_SLOT_HALVES * 2;
    // (stream-ordered behind whatever still reads the old tables; the staging vector is pageable, so the call returns after the copy)
    CHORD_HIP(c, hipMemcpyAsync(c->dShardTables, tab.data(), tab.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, c->stream));
    CHORD_HIP(c, hipStreamSynchronize(c->stream));
    c->shard.ownedRows = c->dShardTables;
    c->shard.tileSlot = reinterpret_cast<const uint32_t*>(c->dShardTables + 64);
    // the owners themselves, one byte per tile: what the sharded cull makes its rank masks from (kernels_cull.hip cluster_rank_mask)
    if (c->dTileOwner) { (void)hipFree(c->dTileOwner); c->dTileOwner = nullptr; }
    CHORD_HIP(c, hipMalloc((void**)&c->dTileOwner, tiles));
    CHORD_HIP(c, hipMemcpy(c->dTileOwner, c->tileOwners.data(), tiles, hipMemcpyHostToDevice));
    c->mineValid = false; c->listMine[1] = c->listMine[2] = false;       // (lists culled for another ownership)
    c->orderAge = c->orderAge1 = 0xFFFFFFFFu;                                            // (a kept tile schedule lists the OLD map's tiles)
    return CHORDVIS_OK;
}

// The exchange buffer of the sharded group cull: ranks x (chunkBlocks x 256 mask words + chunkBlocks triangle sums), chunkBlocks =
// ceil(count blocks / ranks) -- the same on every rank of a frame (same scene, same rank count).
int ensure_cull_exchange(ChordCtx* c)
{
    if (!c->sceneLoaded || c->shard.ranks <= 1 || c->shard.ranks > 8u) return fail(c, CHORDVIS_E_INVALID, "cull exchange: a scene on a context sharded over 2..8 ranks");
    const uint32_t N = c->shard.ranks, chunkBlocks = (c->cullBlocks + N - 1u) / N;
    if (c->dCullExchange && c->cullChunkBlocks == chunkBlocks && c->cullExchangeRanks == N) return CHORDVIS_OK;
    CHORD_HIP(c, hipStreamSynchronize(c->stream));
    if (c->dCullExchange) { (void)hipFree(c->dCullExchange); c->dCullExchange = nullptr; }
    const size_t words = (size_t)N * chunkBlocks * 257u;
    CHORD_HIP(c, hipMalloc((void**)&c->dCullExchange, words * 4u));
    CHORD_HIP(c, hipMemsetAsync(c->dCullExchange, 0, words * 4u, c->stream));
    c->cullChunkBlocks = chunkBlocks; c->cullExchangeRanks = N;
    return CHORDVIS_OK;
}

// The exchange buffer is made when the LAST of {upload_scene, set_shard / allocate_gbuffer} has run, not inside a frame: whether a
// frame starts with the sharded cull -- i.e. whether its first collective happens -- must follow from state every rank shares
// (cull_shardable), never from an allocation that can fail on one rank while its peers are already inside the all-gather.  A failure
// here is the failure of a set-up call.
int prepare_cull_exchange(ChordCtx* c)
{
    if (!c->sceneLoaded || c->shard.ranks <= 1 || c->shard.ranks > 8u) return CHORDVIS_OK;
    return ensure_cull_exchange(c);
}

// Per-context work buffers that depend on the scene's counts (object frames, group masks, command lists, raster work
// lists): the tail of chordvis_upload_scene, shared with the depth-view child context (depth_views.cpp).
int alloc_scene_work_buffers(ChordCtx* c)
{
    int rc;
    const uint64_t instTriangles = c->instTriangles;
    if ((rc = dalloc(c, &c->dObjectsOwned, (size_t)c->objectCount))) return rc;
    if ((rc = dalloc(c, &c->dObjFrame, (size_t)c->objectCount))) return rc;
    if ((rc = dalloc(c, &c->dGroupMask, (size_t)c->groupInstances + 16))) return rc;   // (+16: zeroed in 16-byte vectors)
    if ((rc = dalloc(c, &c->dBlockCounts, (size_t)c->cullBlocks * 3))) return rc;   // counts, triangles, the rank's own counts (sharded), per count block
    c->fullListStale = false;          // (no cull of this scene has written the masks / block counts launch_full_list replays)
    for (int i = 0; i < 3; i++) {
        if ((rc = dalloc(c, &c->lists[i].cmds, (size_t)c->cmdCapacity))) return rc;
        c->lists[i].count = c->dCounts + i;
        c->lists[i].capacity = c->cmdCapacity;
    }
    // raster work lists, sized from the scene like the reference sizes its command buffers from lod0MeshletCount
    // (instance_culling.cpp:141): a triangle instance is set up at most once per frame (stage 0 or stage 1), so the
    // records of a frame never exceed the triangles of every meshlet instance (all LODs) plus the pieces the clipper
    // adds; chordvis_set_limits caps (or, for scenes beyond the default cap, raises) the budget.  Exhaustion is
    // detected on the device and reported by chordvis_stats.  Nearly all triangles take the 32-byte form; the 48-byte
    // list (triangles wider than 64 px, clipped pieces) gets the same bound up to a quarter of the limit.
    // (x2 + 256 Ki: the list is cut into 64 shards that fill unevenly, and a clipped triangle becomes several records)
    const uint64_t need = (2 * instTriangles + (256u << 10) + CHORD_LIST_SHARDS - 1) & ~(uint64_t)(CHORD_LIST_SHARDS - 1);
    c->triCapC = (uint32_t)std::min<uint64_t>(c->limitRecords, need);
    c->triCap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(c->limitRecords / 4, 1u << 20), need) & ~(CHORD_LIST_SHARDS - 1u);
    c->clipTriCap = 1u << 20;
    if ((rc = dalloc(c, &c->dTris, (size_t)c->triCap))) return rc;
    if ((rc = dalloc(c, &c->dTrisC, (size_t)c->triCapC))) return rc;
    // pixel blocks of small clusters: a cluster takes the block form only when that is fewer bytes than its records would
    // be, so the compact list's byte budget bounds the pool too
    c->blockCap = (uint32_t)std::min<uint64_t>((uint64_t)c->triCapC * 2u / CHORD_LIST_SHARDS, CHORD_REC_INDEX_MASK / CHORD_LIST_SHARDS);
    if ((rc = dalloc(c, &c->dBlockPool, (size_t)c->blockCap * CHORD_LIST_SHARDS * 2u))) return rc;
    if ((rc = dalloc(c, &c->dClipTris, (size_t)c->clipTriCap))) return rc;
    c->largeCap = 8u << 20;
    if ((rc = dalloc(c, &c->dLargeList, (size_t)c->largeCap))) return rc;
    return CHORDVIS_OK;
}
}

extern "C" {

// ------------------------------------------------------------------------------------ context --

int chordvis_create(int deviceOrdinal, void* hipStream, ChordCtx** outCtx)
{
    if (!outCtx) return CHORDVIS_E_INVALID;
    *outCtx = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || deviceOrdinal < 0 || deviceOrdinal >= n) return CHORDVIS_E_NO_DEVICE;
    ChordCtx* c = new ChordCtx();
    c->device = deviceOrdinal;
    if (hipSetDevice(deviceOrdinal) != hipSuccess) { delete c; return CHORDVIS_E_NO_DEVICE; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, deviceOrdinal) == hipSuccess) c->numCUs = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (hipStream) { c->stream = (hipStream_t)hipStream; c->ownStream = false; }
    else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return CHORDVIS_E_HIP; }
        c->ownStream = true;
    }
    bool ok = hipMalloc((void**)&c->dTileClocks, sizeof(unsigned long long) * 18 * CHORD_MAX_TILES) == hipSuccess &&
              hipMalloc((void**)&c->dHotTiles, sizeof(uint32_t) * 2 * (1 + CHORD_HOT_TILES)) == hipSuccess &&
              hipMemset(c->dHotTiles, 0, sizeof(uint32_t) * 2 * (1 + CHORD_HOT_TILES)) == hipSuccess &&
              hipMalloc((void**)&c->dView, sizeof(DView)) == hipSuccess &&
              hipMalloc((void**)&c->dFrameState, sizeof(FrameState)) == hipSuccess;
    if (!ok) { chordvis_destroy(c); return CHORDVIS_E_HIP; }
    c->dCounters = &c->dFrameState->counters;
    c->dCounts = c->dFrameState->listCounts;
    (void)hipMemsetAsync(c->dFrameState, 0, sizeof(FrameState), c->stream);
    *outCtx = c;
    return CHORDVIS_OK;
}

int chordvis_destroy(ChordCtx* c)
{
    if (!c) return CHORDVIS_E_INVALID;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    (void)chordvis_comm_destroy(c);
    if (c->depthCtx) { chordvis_destroy(c->depthCtx); c->depthCtx = nullptr; }
    for (float*& d : c->dDepthImages) dfree(d);
    if (c->sharedScene) {       // a depth-view child: the scene buffers are the parent's
        c->dPrims = nullptr; c->dGroups = nullptr; c->dMeshlets = nullptr; c->dGroupIndices = nullptr; c->dMeshletData = nullptr;
        c->dPositions = nullptr; c->dObjStatic = nullptr; c->dGroupRefs = nullptr; c->dMaterials = nullptr; c->dTexAlpha = nullptr;
        c->dTexcoords = nullptr; c->dBvhNodes = nullptr;
    }
    dfree(c->dPrims); dfree(c->dGroups); dfree(c->dMeshlets); dfree(c->dGroupIndices); dfree(c->dMeshletData);
    dfree(c->dPositions); dfree(c->dObjStatic); dfree(c->dMaterials); dfree(c->dTexAlpha); dfree(c->dTexcoords); dfree(c->dBvhNodes); dfree(c->dGroupRefs); dfree(c->dObjectsOwned);
    dfree(c->dView); dfree(c->dObjFrame); dfree(c->dGroupMask); dfree(c->dBlockCounts);
    for (int i = 0; i < 3; i++) dfree(c->lists[i].cmds);
    dfree(c->dRankCmds); dfree(c->dLeftCmds); dfree(c->dMineCmds);
    dfree(c->dCullLookback);
    dfree(c->dFrameState); c->dCounts = nullptr; c->dCounters = nullptr; dfree(c->dTileClocks); dfree(c->dHotTiles); dfree(c->dTileOrder); dfree(c->dTileOrderKeep); dfree(c->dTileOrderKeep1); c->orderAge = c->orderAge1 = 0xFFFFFFFFu; dfree(c->dTileSlabs); dfree(c->dTileMarker); dfree(c->dShadingTiles);
    dfree(c->dVisOwned); dfree(c->dVisResolved);
    for (int i = 0; i < 3; i++) { dfree(c->hzb[i].minTexels); dfree(c->hzb[i].maxTexels); dfree(c->hzb[i].validRange); }
    if (c->hBinHint) { (void)hipHostFree(const_cast<uint32_t*>(c->hBinHint)); c->hBinHint = nullptr; c->dBinHint = nullptr; }
    dfree(c->dRangePartials); dfree(c->dTileRange); dfree(c->dHzbExchange); dfree(c->dHzbFinalExchange); dfree(c->dShardTables); dfree(c->dTileLoads); dfree(c->dTileOwner); dfree(c->dCullExchange); dfree(c->dVisAlt); dfree(c->dVisResolvedAlt); dfree(c->dTris); dfree(c->dTrisC); dfree(c->dBlockPool); dfree(c->dTileBins); dfree(c->dBinPool); dfree(c->dBinChunkTab);
    dfree(c->dClipTris); dfree(c->dLargeList);
    for (hipEvent_t e : c->evPool) (void)hipEventDestroy(e);
    if (c->ownStream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return CHORDVIS_OK;
}

const char* chordvis_last_error(ChordCtx* c) { return c ? c->lastError.c_str() : "null context"; }

int chordvis_sync(ChordCtx* c)
{
    if (!c) return CHORDVIS_E_INVALID;
    { const int rc = flush_pending_tail(c); if (rc) return rc; }
    CHORD_HIP(c, hipStreamSynchronize(c->stream));
    return CHORDVIS_OK;
}

// -------------------------------------------------------------------------------------- scene --

int chordvis_upload_scene(ChordCtx* c, const ChordSceneDesc* s)
{
    if (c) c->orderAge = c->orderAge1 = 0xFFFFFFFFu;                    // (the kept tile schedule of launch_raster is made again)

    if (!c || !s || !s->objects || !s->primitives || !s->materials || !s->assets || s->objectCount == 0)
        return fail(c, CHORDVIS_E_INVALID, "upload_scene: null or empty scene");
    CHORD_HIP(c, hipSetDevice(c->device));

    // asset bases
    std::vector<uint32_t> mB(s->assetCount + 1, 0), gB(s->assetCount + 1, 0), iB(s->assetCount + 1, 0), dB(s->assetCount + 1, 0), vB(s->assetCount + 1, 0);
    for (uint32_t a = 0; a < s->assetCount; a++) {
        const ChordAssetDesc& as = s->assets[a];
        if (!as.meshlets || !as.meshletGroups || !as.meshletGroupIndices || !as.meshletData || !as.positions)
            return fail(c, CHORDVIS_E_INVALID, "upload_scene: asset with null stream");
        mB[a + 1] = mB[a] + as.meshletCount; gB[a + 1] = gB[a] + as.meshletGroupCount;
        iB[a + 1] = iB[a] + as.meshletGroupIndexCount; dB[a + 1] = dB[a] + as.meshletDataCount; vB[a + 1] = vB[a] + as.vertexCount;
    }
    {
        // (the raster kernels form vertex index x 3 in 32 bits -- one full-rate instruction instead of a quarter-rate 64-bit
        // multiply per vertex fetch: 1.43 G vertices = 17 GB of positions per scene; the reference's ByteAddressBuffer offsets
        // are 32-bit BYTE offsets, a third of that)
        uint64_t verts = 0;
        for (uint32_t a = 0; a < s->assetCount; a++) verts += s->assets[a].vertexCount;
        if (verts > 0x55555555ull) return fail(c, CHORDVIS_E_INVALID, "upload_scene: more than 0x55555555 vertices in one scene");
    }
    const uint32_t nM = mB.back(), nG = gB.back(), nI = iB.back(), nD = dB.back(), nV = vB.back();
    std::vector<uint32_t> nB(s->assetCount + 1, 0);
    for (uint32_t a = 0; a < s->assetCount; a++) nB[a + 1] = nB[a] + (s->assets[a].bvhNodes ? s->assets[a].bvhNodeCount : 0u);
    std::vector<DBVHNode> bvh(nB.back());
    bool bvhComplete = nB.back() > 0;

    std::vector<DMeshlet> meshlets(nM);
    std::vector<DGroup> groups(nG);
    std::vector<uint32_t> gidx(nI), mdata(nD);
    std::vector<float> pos((size_t)nV * 3);
    for (uint32_t a = 0; a < s->assetCount; a++) {
        const ChordAssetDesc& as = s->assets[a];
        for (uint32_t i = 0; i < as.meshletCount; i++) {
            const ChordMeshlet& m = as.meshlets[i];
            DMeshlet& d = meshlets[mB[a] + i];
            std::memcpy(&d, &m, sizeof(ChordMeshlet));
            const uint32_t V = m.vertexTriangleCount & 0xFFu, T = (m.vertexTriangleCount >> 8) & 0xFFu;
            if (T > CHORD_MESHLET_MAX_TRIANGLES || (uint64_t)m.dataOffset + V + T > as.meshletDataCount)
                return fail(c, CHORDVIS_E_INVALID, "upload_scene: meshlet exceeds 255 vertices / 128 triangles or its data stream");
            for (uint32_t t = 0; t < T; t++) {                      // local vertex indices of every triangle word
                const uint32_t w = as.meshletData[m.dataOffset + V + t];
                if ((w & 0xFFu) >= V || ((w >> 8) & 0xFFu) >= V || ((w >> 16) & 0xFFu) >= V)
                    return fail(c, CHORDVIS_E_INVALID, "upload_scene: triangle references a vertex beyond the meshlet's vertex count");
            }
            d.dataOffset = m.dataOffset + dB[a];
            d.vertexBase = 0xFFFFFFFFu;
        }
        for (uint32_t i = 0; i < as.meshletGroupCount; i++) {
            const ChordMeshletGroup& g = as.meshletGroups[i];
            DGroup& d = groups[gB[a] + i];
            std::memcpy(&d, &g, sizeof(ChordMeshletGroup));
            d.pad0 = d.pad1 = 0;
            if (g.meshletCount > CHORD_GROUP_MAX_MESHLETS) return fail(c, CHORDVIS_E_INVALID, "upload_scene: group with more than 4 meshlets (nanite_builder.cpp:411-415)");
        }
        std::memcpy(gidx.data() + iB[a], as.meshletGroupIndices, sizeof(uint32_t) * as.meshletGroupIndexCount);
        std::memcpy(mdata.data() + dB[a], as.meshletData, sizeof(uint32_t) * as.meshletDataCount);
        std::memcpy(pos.data() + (size_t)vB[a] * 3, as.positions, sizeof(float) * 3 * as.vertexCount);
    }

    // primitives
    c->hPrims.assign(s->primitiveCount, DPrim{});
    std::vector<uint32_t> primMeshlets(s->primitiveCount, 0);
    std::vector<uint64_t> primTriangles(s->primitiveCount, 0);
    for (uint32_t pi = 0; pi < s->primitiveCount; pi++) {
        const ChordPrimitive& p = s->primitives[pi];
        if (p.primitiveDatasBufferId >= s->assetCount) return fail(c, CHORDVIS_E_INVALID, "upload_scene: primitiveDatasBufferId out of range");
        const uint32_t a = p.primitiveDatasBufferId;
        const ChordAssetDesc& as = s->assets[a];
        if (p.meshletGroupOffset + p.meshletGroupCount > as.meshletGroupCount) return fail(c, CHORDVIS_E_INVALID, "upload_scene: primitive group range out of bounds");
        DPrim& d = c->hPrims[pi];
        std::memcpy(d.posMin, p.posMin, 12); std::memcpy(d.posMax, p.posMax, 12);
        d.meshletBase = mB[a] + p.meshletOffset;
        d.groupBase = gB[a] + p.meshletGroupOffset;
        d.groupIndicesBase = iB[a] + p.meshletGroupIndicesOffset;
        d.groupCount = p.meshletGroupCount;
        d.assetMeshletBase = mB[a];
        d.bvhBase = 0xFFFFFFFFu;
        if (as.bvhNodes && as.bvhNodeCount && p.bvhNodeOffset < as.bvhNodeCount) {
            // GPUBVHNode tree of the primitive (gltf.h:16-24): copied, and checked for what the hierarchical cull relies on --
            // indices in range, every group of the primitive in exactly one node's leaf range, every non-root node's
            // sphere around the parent-error spheres of the groups beneath it
            const ChordBVHNode* nodes = as.bvhNodes + p.bvhNodeOffset;
            const uint32_t count = nodes[0].bvhNodeCount;
            if (count == 0 || (uint64_t)p.bvhNodeOffset + count > as.bvhNodeCount) return fail(c, CHORDVIS_E_INVALID, "upload_scene: BVH root's node count exceeds the node buffer");
            std::vector<uint8_t> seen(p.meshletGroupCount, 0);
            std::vector<std::vector<uint32_t>> below(count);    // parented groups of each node's subtree
            for (uint32_t n = count; n-- > 0;) {            // children come after their parent (breadth-first order)
                const ChordBVHNode& nd = nodes[n];
                if ((uint64_t)nd.leafMeshletGroupOffset + nd.leafMeshletGroupCount > p.meshletGroupCount) return fail(c, CHORDVIS_E_INVALID, "upload_scene: BVH leaf range outside the primitive's groups");
                for (uint32_t g = 0; g < nd.leafMeshletGroupCount; g++) {
                    const uint32_t gi = nd.leafMeshletGroupOffset + g;
                    if (seen[gi]++) return fail(c, CHORDVIS_E_INVALID, "upload_scene: a cluster group is listed by two BVH nodes");
                    const ChordMeshletGroup& gr = as.meshletGroups[p.meshletGroupOffset + gi];
                    if (n != 0) {
                        if (!(gr.parentError < CHORD_ERROR_RADIUS_ROOT)) return fail(c, CHORDVIS_E_INVALID, "upload_scene: an un-parented group below the BVH root");
                        below[n].push_back(gi);
                    }
                }
                for (uint32_t k = 0; k < CHORD_BVH_WIDTH; k++) {
                    const uint32_t ch = nd.children[k];
                    if (ch == CHORD_BVH_NO_CHILD) continue;
                    if (ch <= n || ch >= count) return fail(c, CHORDVIS_E_INVALID, "upload_scene: BVH child index out of order / range");
                    below[n].insert(below[n].end(), below[ch].begin(), below[ch].end());
                    std::vector<uint32_t>().swap(below[ch]);
                }
                // the sphere bounds the parent-error sphere of every group beneath (the root: beneath its children)
                for (uint32_t gi : below[n]) {
                    const ChordMeshletGroup& gr = as.meshletGroups[p.meshletGroupOffset + gi];
                    const float dx = gr.parentPosCenter[0] - nd.sphere[0], dy = gr.parentPosCenter[1] - nd.sphere[1], dz = gr.parentPosCenter[2] - nd.sphere[2];
                    if (!(std::sqrt(dx * dx + dy * dy + dz * dz) + gr.parentError <= nd.sphere[3] * 1.0001f + 1.0e-6f))
                        return fail(c, CHORDVIS_E_INVALID, "upload_scene: a BVH node's sphere does not contain the parent-error spheres beneath it");
                }
                DBVHNode& dn = bvh[nB[a] + p.bvhNodeOffset + n];
                std::memcpy(dn.sphere, nd.sphere, 16); std::memcpy(dn.children, nd.children, 32);
                dn.bvhNodeCount = nd.bvhNodeCount; dn.leafGroupOffset = nd.leafMeshletGroupOffset; dn.leafGroupCount = nd.leafMeshletGroupCount; dn.pad = 0;
            }
            for (uint32_t gi = 0; gi < p.meshletGroupCount; gi++) if (!seen[gi]) return fail(c, CHORDVIS_E_INVALID, "upload_scene: a cluster group is missing from the primitive's BVH");
            {   // breadth-first order: a level is a contiguous range; every node keeps the end of its level (DBVHNode::pad)
                std::vector<uint32_t> level(count, 0xFFFFFFFFu);
                level[0] = 0;
                for (uint32_t n = 0; n < count; n++) {
                    if (level[n] == 0xFFFFFFFFu) return fail(c, CHORDVIS_E_INVALID, "upload_scene: a BVH node is not reachable from the root");
                    for (uint32_t k = 0; k < CHORD_BVH_WIDTH; k++) {
                        const uint32_t ch = nodes[n].children[k];
                        if (ch == CHORD_BVH_NO_CHILD) continue;
                        if (level[ch] != 0xFFFFFFFFu) return fail(c, CHORDVIS_E_INVALID, "upload_scene: a BVH node has two parents");
                        level[ch] = level[n] + 1;
                    }
                    if (n && level[n] < level[n - 1]) return fail(c, CHORDVIS_E_INVALID, "upload_scene: BVH nodes are not in breadth-first order");
                    if (level[n] >= CHORD_BVH_MAX_LEVELS) return fail(c, CHORDVIS_E_INVALID, "upload_scene: BVH deeper than kNaniteMaxBVHLevelCount");
                }
                uint32_t end = count;
                for (uint32_t n = count; n-- > 0;) {
                    if (n + 1 < count && level[n + 1] != level[n]) end = n + 1;
                    bvh[nB[a] + p.bvhNodeOffset + n].pad = end;
                }
            }
            d.bvhBase = nB[a] + p.bvhNodeOffset;
        } else bvhComplete = false;
        for (uint32_t gi = 0; gi < p.meshletGroupCount; gi++) {
            const ChordMeshletGroup& g = as.meshletGroups[p.meshletGroupOffset + gi];
            for (uint32_t i = 0; i < g.meshletCount; i++) {
                const uint32_t ii = p.meshletGroupIndicesOffset + g.meshletOffset + i;
                if (ii >= as.meshletGroupIndexCount) return fail(c, CHORDVIS_E_INVALID, "upload_scene: group index out of bounds");
                const uint32_t mi = p.meshletOffset + as.meshletGroupIndices[ii];
                if (mi >= as.meshletCount) return fail(c, CHORDVIS_E_INVALID, "upload_scene: meshlet index out of bounds");
                DMeshlet& dm = meshlets[mB[a] + mi];
                if (dm.vertexBase == 0xFFFFFFFFu) {                  // first reference: check the vertex ids against the asset
                    const ChordMeshlet& sm = as.meshlets[mi];
                    const uint32_t V = sm.vertexTriangleCount & 0xFFu;
                    for (uint32_t v = 0; v < V; v++)
                        if ((uint64_t)p.vertexOffset + as.meshletData[sm.dataOffset + v] >= as.vertexCount)
                            return fail(c, CHORDVIS_E_INVALID, "upload_scene: meshlet vertex id out of the asset's position stream");
                }
                dm.vertexBase = vB[a] + p.vertexOffset;
                primMeshlets[pi]++;
                primTriangles[pi] += (as.meshlets[mi].vertexTriangleCount >> 8) & 0xFFu;
            }
        }
    }

    // objects
    c->hObjStatic.assign(s->objectCount, DObjStatic{});
    uint64_t groupInst = 0, cmdCap = 0, instTriangles = 0;
    for (uint32_t o = 0; o < s->objectCount; o++) {
        const ChordObject& ob = s->objects[o];
        if (ob.GLTFPrimitiveDetail >= s->primitiveCount || ob.GLTFMaterialData >= s->materialCount)
            return fail(c, CHORDVIS_E_INVALID, "upload_scene: object primitive/material id out of range");
        DObjStatic& d = c->hObjStatic[o];
        d.prim = ob.GLTFPrimitiveDetail;
        const ChordMaterial& mat = s->materials[ob.GLTFMaterialData];
        // alphaMode 0 / 1 / 2 = opaque / mask / blend (FILTER_CHECK of pipeline_filter.hlsl:100-101); anything else never
        // matches a bucket's targetAlphaMode and draws nothing, like blend
        d.matFlags = (mat.bTwoSided != 0 ? CHORD_MATFLAG_TWO_SIDED : 0u) | ((mat.alphaMode > 2u ? 2u : mat.alphaMode) << 1) | (ob.GLTFMaterialData << 8);
        if (ob.GLTFMaterialData >= (1u << 24)) return fail(c, CHORDVIS_E_CAPACITY, "upload_scene: more than 2^24 materials");
        d.shadingType = s->materials[ob.GLTFMaterialData].materialType;
        d.groupBase = (uint32_t)groupInst;
        for (int i = 0; i < 3; i++) { d.posMin[i] = c->hPrims[d.prim].posMin[i]; d.posMax[i] = c->hPrims[d.prim].posMax[i]; }
        groupInst += c->hPrims[d.prim].groupCount;
        cmdCap += primMeshlets[d.prim];
        instTriangles += primTriangles[d.prim];
    }
    if (cmdCap >= CHORD_MAX_INSTANCE_ID || groupInst > 0x7FFFFFFFull)
        return fail(c, CHORDVIS_E_CAPACITY, "upload_scene: more than 2^24-2 cluster instances do not fit the 24-bit visibility id (base.h:412)");
    // the flattened (object, group) instances, every indirection of the group cull resolved (DGroupRef)
    std::vector<DGroupRef> refs((size_t)groupInst);
    for (uint32_t o = 0; o < s->objectCount; o++) {
        const DObjStatic& d = c->hObjStatic[o];
        const DPrim& pr = c->hPrims[d.prim];
        for (uint32_t gl = 0; gl < pr.groupCount; gl++) {
            const DGroup& g = groups[pr.groupBase + gl];
            DGroupRef& r = refs[(size_t)d.groupBase + gl];
            const uint32_t cnt = std::min(g.meshletCount, (uint32_t)CHORD_GROUP_MAX_MESHLETS);
            r.object = o;
            r.group = (pr.groupBase + gl) | (cnt << 28);
            for (uint32_t i = 0; i < CHORD_GROUP_MAX_MESHLETS; i++)
                r.meshlet[i] = cnt ? pr.meshletBase + gidx[pr.groupIndicesBase + g.meshletOffset + (i < cnt ? i : 0u)] : 0u;
        }
    }
    if (nG >= (1u << 28)) return fail(c, CHORDVIS_E_CAPACITY, "upload_scene: more than 2^28 cluster groups");

    dfree(c->dRankCmds);                                   // sized by cmdCapacity; re-made by the first sharded raster pass
    dfree(c->dLeftCmds); dfree(c->dMineCmds);
    c->mineValid = false; c->listMine[1] = c->listMine[2] = false;       // (the rank's lists went with the buffers)
    c->fullListStale = false; c->cullPhaseDone = false;                    // (the masks of the old scene's last cull are gone: nothing to replay)
    dfree(c->dCullExchange);                                               // (sized by the scene's count blocks: re-made on demand)
    c->objectCount = s->objectCount; c->primCount = s->primitiveCount; c->materialCount = s->materialCount;
    c->meshletCount = nM; c->groupCount = nG;
    c->groupInstances = (uint32_t)groupInst; c->cmdCapacity = (uint32_t)std::max<uint64_t>(cmdCap, 1);
    c->cullBlocks = std::max(1u, (c->groupInstances + 255u) / 256u);

    // what the masked buckets sample: alpha channels of every texture level, materials with texture + sampler resolved,
    // the per-vertex texture coordinates (mesh_raster.hlsl:107-112,198-204)
    std::vector<DMaterial> dmats(s->materialCount);
    std::vector<uint8_t> alpha;
    std::vector<uint32_t> texOffset(s->textureCount, 0xFFFFFFFFu);
    bool anyMasked = false;
    for (uint32_t m = 0; m < s->materialCount; m++) anyMasked = anyMasked || s->materials[m].alphaMode == CHORD_ALPHA_MASK;
    if (anyMasked && s->textures) {
        // only the textures an alpha-tested material samples are looked at (a host may hand over its whole bindless table, with
        // pixel data only where the visibility pass needs it); every other entry stays "white"
        std::vector<uint8_t> sampled(s->textureCount, 0);
        for (uint32_t m = 0; m < s->materialCount; m++)
            if (s->materials[m].alphaMode == CHORD_ALPHA_MASK && s->materials[m].baseColorId < s->textureCount) sampled[s->materials[m].baseColorId] = 1;
        for (uint32_t t = 0; t < s->textureCount; t++) {
            if (!sampled[t]) continue;
            const ChordTexture& tx = s->textures[t];
            if (!tx.rgba8 || tx.width == 0 || tx.height == 0 || tx.mipCount == 0 || tx.width > 16384u || tx.height > 16384u || tx.mipCount > 15u)
                return fail(c, CHORDVIS_E_INVALID, "upload_scene: texture without data, or larger than 16384 / 15 levels");
            size_t texels = 0;
            for (uint32_t l = 0; l < tx.mipCount; l++) texels += (size_t)std::max(1u, tx.width >> l) * std::max(1u, tx.height >> l);
            if (alpha.size() + texels >= 0xFFFFFFFFull) return fail(c, CHORDVIS_E_CAPACITY, "upload_scene: more than 4 G texels of alpha");
            texOffset[t] = (uint32_t)alpha.size();
            const size_t base = alpha.size();
            alpha.resize(base + texels);
            for (size_t i = 0; i < texels; i++) alpha[base + i] = tx.rgba8[i * 4 + 3];
        }
    }
    for (uint32_t m = 0; m < s->materialCount; m++) {
        const ChordMaterial& mat = s->materials[m];
        DMaterial& d = dmats[m];
        std::memset(&d, 0, sizeof(d));
        d.texOffset = 0xFFFFFFFFu;
        if (s->textures && mat.baseColorId < s->textureCount && texOffset[mat.baseColorId] != 0xFFFFFFFFu) {
            const ChordTexture& tx = s->textures[mat.baseColorId];
            d.texOffset = texOffset[mat.baseColorId]; d.texWidth = tx.width; d.texHeight = tx.height; d.texMips = tx.mipCount;
        }
        if (s->samplers && mat.baseColorSampler < s->samplerCount) {
            const ChordSampler& sm = s->samplers[mat.baseColorSampler];
            d.minFilter = sm.minFilter; d.magFilter = sm.magFilter; d.wrapS = sm.wrapS; d.wrapT = sm.wrapT;
        } else { d.minFilter = d.magFilter = CHORD_FILTER_NEAREST; d.wrapS = d.wrapT = CHORD_WRAP_REPEAT; }
        d.alphaFactor = mat.baseColorFactor[3]; d.alphaCutOff = mat.alphaCutOff;
        // every level as the tile kernel's row units want it (DMatLevel)
        if (d.texOffset != 0xFFFFFFFFu) {
            uint32_t off = d.texOffset;
            auto wrap_consts = [](uint32_t n, uint32_t mode, uint32_t& magic, uint32_t& bias) {
                magic = 0u; bias = 0u;
                if (mode == CHORD_WRAP_CLAMP_TO_EDGE) return;
                const uint32_t period = mode == CHORD_WRAP_MIRRORED_REPEAT ? 2u * n : n;
                if ((period & (period - 1u)) == 0u) return;                  // masked, not divided
                magic = (uint32_t)(0x100000000ull / period);
                bias = (uint32_t)(((0x40000000ull + period - 1u) / period) * period);
            };
            for (uint32_t l = 0; l < d.texMips && l < CHORD_MAX_TEX_LEVELS; l++) {
                const uint32_t w = std::max(1u, d.texWidth >> l), h = std::max(1u, d.texHeight >> l);
                chord::DMatLevel& L = d.levels[l];
                L.base = off; L.dims = (w - 1u) | (h - 1u) << 16;
                wrap_consts(w, d.wrapS, L.magicS, L.biasS);
                wrap_consts(h, d.wrapT, L.magicT, L.biasT);
                off += w * h;
            }
        }
    }
    std::vector<float> uvs;
    if (anyMasked) {
        bool anyUv = false;
        for (uint32_t a = 0; a < s->assetCount; a++) anyUv = anyUv || (s->assets[a].texcoord0 && s->assets[a].texcoord0Count);
        if (anyUv) {
            uvs.assign((size_t)nV * 2, 0.0f);
            for (uint32_t a = 0; a < s->assetCount; a++) {
                const ChordAssetDesc& as = s->assets[a];
                if (as.texcoord0 && as.texcoord0Count) {
                    // (a shorter array would silently read as uv = (0, 0) for the vertices beyond it)
                    if (as.texcoord0Count != as.vertexCount) return fail(c, CHORDVIS_E_INVALID, "upload_scene: texcoord0Count must equal vertexCount (or be 0: no texture coordinates)");
                    std::memcpy(uvs.data() + (size_t)vB[a] * 2, as.texcoord0, sizeof(float) * 2 * as.vertexCount);
                }
            }
        }
    }
    c->anyMasked = anyMasked;

    int rc;
#define UP(dst, vec)                                                                                          \
    if ((rc = dalloc(c, &dst, vec.size()))) return rc;                                                        \
    if (!vec.empty()) CHORD_HIP(c, hipMemcpy(dst, vec.data(), vec.size() * sizeof(vec[0]), hipMemcpyHostToDevice));
    UP(c->dMeshlets, meshlets) UP(c->dGroups, groups) UP(c->dGroupIndices, gidx) UP(c->dMeshletData, mdata)
    UP(c->dPositions, pos) UP(c->dPrims, c->hPrims) UP(c->dObjStatic, c->hObjStatic) UP(c->dGroupRefs, refs)
    UP(c->dMaterials, dmats)
    if (!bvh.empty()) { UP(c->dBvhNodes, bvh) } else dfree(c->dBvhNodes);
    c->bvhComplete = bvhComplete;
    if (!alpha.empty()) { UP(c->dTexAlpha, alpha) } else dfree(c->dTexAlpha);
    if (!uvs.empty()) { UP(c->dTexcoords, uvs) } else dfree(c->dTexcoords);
#undef UP
    c->instTriangles = instTriangles;
    if (c->depthCtx) { chordvis_destroy(c->depthCtx); c->depthCtx = nullptr; }       // (it aliased the old scene buffers)
    if ((rc = chord::alloc_scene_work_buffers(c))) return rc;
    CHORD_HIP(c, hipMemcpy(c->dObjectsOwned, s->objects, sizeof(ChordObject) * s->objectCount, hipMemcpyHostToDevice));
    c->dObjects = c->dObjectsOwned;
    c->sceneLoaded = true;
    c->historySlot = 0;
    c->pendingTailSlot = 0;
    if ((rc = chord::prepare_cull_exchange(c))) { c->sceneLoaded = false; return rc; }     // (a sharded context: the rank-mask exchange buffer of this scene)
    if (c->cullMode == 1 && !c->bvhComplete) return fail(c, CHORDVIS_E_INVALID, "upload_scene: hierarchical culling is selected and a primitive has no BVH");
    return CHORDVIS_OK;
}

int chordvis_update_objects(ChordCtx* c, const ChordObject* hostObjects, uint32_t count)
{
    if (!c || !c->sceneLoaded || !hostObjects || count != c->objectCount) return fail(c, CHORDVIS_E_INVALID, "update_objects: count must equal the uploaded scene's objectCount");
    CHORD_HIP(c, hipMemcpyAsync(c->dObjectsOwned, hostObjects, sizeof(ChordObject) * count, hipMemcpyHostToDevice, c->stream));
    c->dObjects = c->dObjectsOwned;
    return CHORDVIS_OK;
}

int chordvis_bind_objects(ChordCtx* c, const ChordObject* deviceObjects, uint32_t count)
{
    if (!c || !c->sceneLoaded || !deviceObjects || count != c->objectCount) return fail(c, CHORDVIS_E_INVALID, "bind_objects: count must equal the uploaded scene's objectCount");
    c->dObjects = deviceObjects;
    return CHORDVIS_OK;
}

int chordvis_set_view(ChordCtx* c, const ChordCameraView* view, const ChordInstanceCullingView* iv, uint32_t switchFlags)
{
    if (!c || !view || !iv) return fail(c, CHORDVIS_E_INVALID, "set_view: null argument");
    // (the rank masks on their way through the all-gather were computed with the current view, cull mode and tile map)
    if (c->cullPhaseDone) return fail(c, CHORDVIS_E_INVALID, "set_view: not between chordvis_frame_phase_cull and chordvis_frame_phase_a");
    c->hView.view = *view; c->hView.iv = *iv; c->hView.flags = switchFlags;
    c->hView.width = (uint32_t)iv->renderDimension[0]; c->hView.height = (uint32_t)iv->renderDimension[1];
    if (c->dVis && (c->hView.width != c->width || c->hView.height != c->height))
        return fail(c, CHORDVIS_E_INVALID, "set_view: renderDimension differs from the allocated gbuffer");
    // the 600-byte constant block travels as a kernel argument of the frame's first kernel (object_cull),
    // which publishes it for the later passes; see flush_view() for passes called out of frame order
    c->viewDirty = true;
    c->viewSet = true;
    return CHORDVIS_OK;
}

int chordvis_allocate_gbuffer(ChordCtx* c, uint32_t width, uint32_t height, uint64_t* deviceVisibility)
{
    if (!c || width < 64 || height < 64 || width > 4096 || height > 4096)      // renderer.h:52-53
        return fail(c, CHORDVIS_E_INVALID, "allocate_gbuffer: render size must be 64..4096 per axis (renderer.h:52-53)");
    CHORD_HIP(c, hipSetDevice(c->device));
    c->width = width; c->height = height;
    dfree(c->dTileMarker); dfree(c->dShadingTiles);                             // sized by the render size; re-made on demand
    return configure_targets(c, deviceVisibility);
}

int chordvis_set_limits(ChordCtx* c, const ChordLimits* limits)
{
    if (c) c->orderAge = c->orderAge1 = 0xFFFFFFFFu;                    // (the kept tile schedule of launch_raster is made again)

    if (!c || !limits) return fail(c, CHORDVIS_E_INVALID, "set_limits: null argument");
    if (c->sceneLoaded || c->dVis) return fail(c, CHORDVIS_E_INVALID, "set_limits: call before upload_scene / allocate_gbuffer");
    if (limits->maxTriangleRecords) {
        if (limits->maxTriangleRecords < (1u << 16) || limits->maxTriangleRecords > 0x7FFFFFC0ull)
            return fail(c, CHORDVIS_E_INVALID, "set_limits: maxTriangleRecords out of range (record indices are 32-bit)");
        c->limitRecords = limits->maxTriangleRecords & ~(uint64_t)(CHORD_LIST_SHARDS - 1);
    }
    if (limits->binPoolChunks) {
        if (limits->binPoolChunks > (16u << 20)) return fail(c, CHORDVIS_E_INVALID, "set_limits: at most 16 Mi pool chunks per pass (64 GB)");
        c->limitPoolChunks = limits->binPoolChunks;
    }
    if (limits->binMaxChunksPerTile) {
        if (limits->binMaxChunksPerTile > CHORD_BIN_MAX_CHUNKS_LIMIT) return fail(c, CHORDVIS_E_INVALID, "set_limits: at most 3072 overflow chunks per tile");
        c->binMaxChunks = limits->binMaxChunksPerTile;
    }
    return CHORDVIS_OK;
}

int chordvis_set_tile_schedule_keep(ChordCtx* c, uint32_t frames)
{
    if (!c) return CHORDVIS_E_INVALID;
    c->orderKeepFrames = frames;
    c->orderAge = c->orderAge1 = 0xFFFFFFFFu;
    return CHORDVIS_OK;
}
uint32_t chordvis_tile_schedule_keep(ChordCtx* c) { return c ? c->orderKeepFrames : 0u; }

int chordvis_set_cull_mode(ChordCtx* c, int hierarchical)
{
    if (!c || hierarchical < 0 || hierarchical > 1) return fail(c, CHORDVIS_E_INVALID, "set_cull_mode: 0 (flat) or 1 (hierarchical)");
    if (hierarchical && c->sceneLoaded && !c->bvhComplete) return fail(c, CHORDVIS_E_INVALID, "set_cull_mode: the uploaded scene has primitives without a BVH");
    if (c->cullPhaseDone) return fail(c, CHORDVIS_E_INVALID, "set_cull_mode: not between chordvis_frame_phase_cull and chordvis_frame_phase_a");
    c->cullMode = hierarchical;
    return CHORDVIS_OK;
}

int chordvis_set_shard(ChordCtx* c, uint32_t ranks, uint32_t rank)
{
    if (c) c->orderAge = c->orderAge1 = 0xFFFFFFFFu;                    // (the kept tile schedule of launch_raster is made again)

    if (!c || ranks == 0 || ranks > 255u || rank >= ranks) return fail(c, CHORDVIS_E_INVALID, "set_shard: rank < ranks <= 255");
#if CHORD_TILE_SHIFT != 6
    if (ranks > 1) return fail(c, CHORDVIS_E_INVALID, "set_shard: this build's raster tiles are not 64 x 64");
#endif
    if (c->inFrame || c->cullPhaseDone) return fail(c, CHORDVIS_E_INVALID, "set_shard: not inside a frame");
    if (c->shard.ranks != ranks) { c->tileOwners.clear(); c->tileOwnersExplicit = false; }
    c->shard.ranks = ranks; c->shard.rank = rank;
    c->mineValid = false; c->listMine[1] = c->listMine[2] = false;       // (lists culled for another ownership)
    if (c->width) {
        if (c->visExternal) { c->dVis = nullptr; return fail(c, CHORDVIS_E_INVALID, "set_shard after allocate_gbuffer with an external buffer: call allocate_gbuffer again"); }
        return configure_targets(c, nullptr);
    }
    return CHORDVIS_OK;
}

// An explicit tile map (one owner per tile, row-major over the tile grid; the same table on every rank), e.g. from
// chordvis_tile_layout with the loads of a rendered frame.  Between frames; the history HZB carries over (it is not sharded).
int chordvis_set_tile_owners(ChordCtx* c, const uint8_t* owners, uint32_t tiles)
{
    if (!c || c->shard.ranks <= 1 || !c->width) return fail(c, CHORDVIS_E_INVALID, "set_tile_owners: a sharded context with a G-buffer");
    std::vector<uint8_t> def;
    if (!owners) {                                                        // NULL: back to the default map
        tiles = c->tilesX * c->tilesY;
        def.assign(tiles, 0);
        if (tile_layout(c->tilesX, c->tilesY, c->shard.ranks, nullptr, 0u, def.data()) != CHORDVIS_OK) return fail(c, CHORDVIS_E_INVALID, "tile layout");
        owners = def.data();
    }
    if (tiles != c->tilesX * c->tilesY) return fail(c, CHORDVIS_E_INVALID, "set_tile_owners: one owner per tile of the G-buffer");
    if (c->inFrame || c->cullPhaseDone) return fail(c, CHORDVIS_E_INVALID, "set_tile_owners: not inside a frame");
    // frames in flight (pipelined hosts: an image still travelling) were laid out with the old map
    CHORD_HIP(c, hipStreamSynchronize(c->stream));
    if (c->commResolveStream) CHORD_HIP(c, hipStreamSynchronize(c->commResolveStream));
    for (int k = 0; k < 2; k++) if (c->visReadyEvent[k]) CHORD_HIP(c, hipEventSynchronize(c->visReadyEvent[k]));
    std::vector<uint8_t> keep = c->tileOwners;
    c->tileOwners.assign(owners, owners + tiles);
    const int rc = install_tile_owners(c);
    if (rc) { c->tileOwners = keep; if (!keep.empty()) (void)install_tile_owners(c); return rc; }
    c->tileOwnersExplicit = def.empty();
    return CHORDVIS_OK;
}

int chordvis_get_tile_owners(ChordCtx* c, uint8_t* ownersOut, uint32_t tiles)
{
    if (!c || !ownersOut || c->shard.ranks <= 1 || c->tileOwners.size() != tiles) return fail(c, CHORDVIS_E_INVALID, "get_tile_owners: a sharded context with a G-buffer, one entry per tile");
    std::memcpy(ownersOut, c->tileOwners.data(), tiles);
    return CHORDVIS_OK;
}

// Bin entries per tile of the last frame this context finished, every rank's tiles (they travel in the end-of-frame exchange).
// Waits for the context's stream.
int chordvis_read_tile_loads(ChordCtx* c, uint32_t* loadsOut, uint32_t tiles)
{
    if (!c || !loadsOut || c->shard.ranks <= 1 || !c->dTileLoads || tiles != c->tilesX * c->tilesY) return fail(c, CHORDVIS_E_INVALID, "read_tile_loads: a sharded context with a G-buffer, one entry per tile");
    CHORD_HIP(c, hipStreamSynchronize(c->stream));
    CHORD_HIP(c, hipMemcpy(loadsOut, c->dTileLoads, sizeof(uint32_t) * tiles, hipMemcpyDeviceToHost));
    return CHORDVIS_OK;
}

// Re-balances the tile map from the last frame's loads: chordvis_read_tile_loads -> chordvis_tile_layout -> chordvis_set_tile_owners.
// Every rank of the frame must call it at the same frame boundary (they hold the same loads, so they compute the same map).
// imbalance (may be NULL): max over ranks of the OLD map's load / the mean, x 1000.
int chordvis_rebalance(ChordCtx* c, uint32_t* imbalancePermille)
{
    if (!c || c->shard.ranks <= 1 || !c->dTileLoads) return fail(c, CHORDVIS_E_INVALID, "rebalance: a sharded context with a G-buffer");
    const uint32_t tiles = c->tilesX * c->tilesY, N = c->shard.ranks;
    std::vector<uint32_t> loads(tiles);
    int rc = chordvis_read_tile_loads(c, loads.data(), tiles);
    if (rc) return rc;
    std::vector<uint64_t> per(N, 0);
    uint64_t total = 0;
    for (uint32_t t = 0; t < tiles; t++) { per[c->tileOwners[t]] += loads[t]; total += loads[t]; }
    if (imbalancePermille) *imbalancePermille = total ? (uint32_t)(*std::max_element(per.begin(), per.end()) * N * 1000u / total) : 1000u;
    if (total == 0) return CHORDVIS_OK;                                  // (no frame rendered yet: keep the map)
    std::vector<uint8_t> owners(tiles);
    if (tile_layout(c->tilesX, c->tilesY, N, loads.data(), c->slotCapacity, owners.data()) != CHORDVIS_OK) return fail(c, CHORDVIS_E_INVALID, "rebalance: tile layout");
    return chordvis_set_tile_owners(c, owners.data(), tiles);
}

uint64_t chordvis_visibility_words(ChordCtx* c)
{
    if (!c) return 0;
    if (c->width == 0) return 0;
    const uint32_t N = c->shard.ranks;
    return N > 1 ? (uint64_t)N * chordvis_tile_slot_capacity(c->width, c->height, N) * (CHORD_TILE * CHORD_TILE) : (uint64_t)c->width * c->height;
}
uint64_t chordvis_visibility_chunk_words(ChordCtx* c) { return !c ? 0 : c->shard.ranks > 1 ? (uint64_t)c->shard.slotsPerRank * (CHORD_TILE * CHORD_TILE) : chordvis_visibility_words(c); }
uint64_t* chordvis_visibility_ptr(ChordCtx* c) { return c ? c->dVis : nullptr; }
uint64_t* chordvis_resolved_visibility_ptr(ChordCtx* c) { return c ? (c->shard.ranks > 1 ? c->dVisResolved : c->dVis) : nullptr; }
uint16_t* chordvis_hzb_exchange_ptr(ChordCtx* c) { return c ? c->dHzbExchange : nullptr; }
uint16_t* chordvis_hzb_final_exchange_ptr(ChordCtx* c) { return c ? c->dHzbFinalExchange : nullptr; }
uint64_t chordvis_hzb_final_exchange_chunk_bytes(ChordCtx* c) { return c ? c->hzbFinalExchangeChunkBytes : 0; }

// Pipelined sharded frames keep TWO frames' visibility words alive: the one whose all-gather is still travelling and the
// one being rasterized.  Swaps the roles of the two buffer pairs (the second pair is allocated on first use); call between
// frames, before chordvis_frame_phase_a.
int chordvis_swap_visibility(ChordCtx* c)
{
    if (!c || !c->dVis) return fail(c, CHORDVIS_E_INVALID, "swap_visibility: allocate_gbuffer must come first");
    if (c->shard.ranks <= 1 || c->visExternal) return fail(c, CHORDVIS_E_INVALID, "swap_visibility: only for sharded contexts that own their visibility buffer");
    if (c->inFrame) return fail(c, CHORDVIS_E_INVALID, "swap_visibility: not inside a frame");
    int rc;
    if (!c->dVisAlt) {
        if ((rc = dalloc(c, &c->dVisAlt, c->visWords))) return rc;
        if ((rc = dalloc(c, &c->dVisResolvedAlt, (uint64_t)c->width * c->height))) return rc;
        CHORD_HIP(c, hipMemsetAsync(c->dVisAlt, 0, c->visWords * 8, c->stream));   // (incl. the padding of edge tiles' slots)
        CHORD_HIP(c, hipMemsetAsync(c->dVisResolvedAlt, 0, (uint64_t)c->width * c->height * 8, c->stream));
    }
    std::swap(c->dVisOwned, c->dVisAlt);
    c->dVis = c->dVisOwned;
    std::swap(c->dVisResolved, c->dVisResolvedAlt);
    std::swap(c->visReadyEvent[0], c->visReadyEvent[1]);
    return CHORDVIS_OK;
}
uint64_t chordvis_hzb_exchange_halves(ChordCtx* c) { return c ? c->hzbExchangeHalves : 0; }
// the sharded cull's exchange buffer (made on first request, after upload_scene and chordvis_set_shard; NULL / 0 when the sharded
// cull does not apply: one rank, more than 8, no scene)
uint32_t* chordvis_cull_exchange_ptr(ChordCtx* c) { return (c && c->sceneLoaded && c->shard.ranks > 1 && c->shard.ranks <= 8u && ensure_cull_exchange(c) == CHORDVIS_OK) ? c->dCullExchange : nullptr; }
uint64_t chordvis_cull_exchange_chunk_bytes(ChordCtx* c) { return chordvis_cull_exchange_ptr(c) ? (uint64_t)c->cullChunkBlocks * 257u * 4u : 0; }
// measurement / test aid: fills EVERY rank's chunk of the cull exchange buffer on this context (what the all-gather would deliver),
// for the current view -- tools/shard_time.py times one rank at a time on one device
int chordvis_debug_fill_cull_exchange(ChordCtx* c)
{
    int rc = ready(c, "debug_fill_cull_exchange");
    if (rc) return rc;
    if (!cull_shardable_config(c)) return fail(c, CHORDVIS_E_INVALID, "debug_fill_cull_exchange: the sharded cull does not apply");
    if ((rc = ensure_cull_exchange(c))) return rc;
    if (c->inFrame) return fail(c, CHORDVIS_E_INVALID, "debug_fill_cull_exchange: not inside a frame");
    if ((rc = flush_view(c))) return rc;
    launch_cull_masks(c, true);
    CHORD_HIP(c, hipGetLastError());
    return CHORDVIS_OK;
}
uint64_t chordvis_hzb_exchange_chunk_halves(ChordCtx* c) { return c ? c->hzbExchangeChunkHalves : 0; }

// -------------------------------------------------------------------------------------- passes --

int chordvis_clear_gbuffer(ChordCtx* c)
{
    int rc = ready(c, "clear_gbuffer");
    if (rc) return rc;
    // visibility = 0 / depth = 0.0 (render_textures.cpp:81-85,98-100).  Sharded: only the rank's own
    // chunk needs clearing (the all-gather overwrites the rest).
    if (c->shard.ranks > 1) {
        const uint64_t chunk = (uint64_t)c->shard.slotsPerRank * (CHORD_TILE * CHORD_TILE);
        CHORD_HIP(c, hipMemsetAsync(c->dVis + chunk * c->shard.rank, 0, chunk * 8, c->stream));
    } else {
        CHORD_HIP(c, hipMemsetAsync(c->dVis, 0, c->visWords * 8, c->stream));
    }
    CHORD_HIP(c, hipMemsetAsync(c->dCounters, 0, sizeof(DeviceCounters), c->stream));
    c->rasterCalls = 0;
    c->pendingClear = false;
    c->inFrame = false;
    return CHORDVIS_OK;
}

int chordvis_instance_culling(ChordCtx* c, ChordCountAndCmd* out)
{
    int rc = ready(c, "instance_culling");
    if (rc) return rc;
    launch_group_cull(c, c->lists[0]);                   // instanceCullingCS + clusterGroupCullingCS (objects + count, scatter)
    CHORD_HIP(c, hipGetLastError());
    if (out) *out = c->lists[0].handle();
    return CHORDVIS_OK;
}

int chordvis_hzb_culling(ChordCtx* c, const ChordHZB* hzb, int bFirstStage, ChordCountAndCmd in,
                         ChordCountAndCmd* outVisible, ChordCountAndCmd* outRejected)
{
    int rc = ready(c, "hzb_culling");
    if (rc) return rc;
    if (!hzb || !hzb->minTexels || !in.count || !in.cmds) return fail(c, CHORDVIS_E_INVALID, "hzb_culling: invalid HZB or command list");
    if ((rc = flush_pending_tail(c))) return rc;
    if ((rc = flush_view(c))) return rc;
    const HzbBuffers hb = from_handle(hzb);
    CmdList inL = from_handle(in);
    // Sharded frames: a rank culls (and later rasters) only the clusters that touch its pixel rows -- the group cull wrote them
    // to a list of their own -- so the occlusion tests shard with the screen like the raster does.  The decision per cluster
    // is a function of the cluster and the (exchanged, identical) HZB, so every owner of a straddling cluster decides alike.
    bool inMine = false;
    if (c->shard.ranks > 1) {
        if (inL.cmds == c->lists[0].cmds && c->mineValid) { inL.count = c->dCounts + 4; inL.cmds = c->dMineCmds; inMine = true; }
        else if (inL.cmds == c->dMineCmds && c->mineValid) inMine = true;
        else for (int k = 1; k < 3; k++) if (inL.cmds == c->lists[k].cmds && c->listMine[k]) inMine = true;
    }
    if (bFirstStage) {
        if (inL.cmds == c->lists[1].cmds || inL.cmds == c->lists[2].cmds) return fail(c, CHORDVIS_E_INVALID, "hzb_culling: first stage input must be the instanceCulling list");
        if (c->fusedCullDone && inL.cmds == c->lists[0].cmds && hzb->minTexels == c->hzb[c->historySlot].minTexels) {
            // (chordvis_render_frame on a short scene: frame_cull_fused_kernel tested every command it emitted against this chain)
            c->fusedCullDone = false;
        } else {
        if (!c->inFrame) CHORD_HIP(c, hipMemsetAsync(c->dCounts + 1, 0, 8, c->stream));
        c->listMine[1] = c->listMine[2] = inMine;
        launch_hzb_cull(c, hb, 0, inL, c->lists[1], &c->lists[2]);
        }
        if (outVisible) *outVisible = c->lists[1].handle();
        if (outRejected) *outRejected = c->lists[2].handle();
    } else {
        if (inL.cmds == c->lists[1].cmds) return fail(c, CHORDVIS_E_INVALID, "hzb_culling: second stage input aliases its output");
        CmdList vis1 = c->lists[1];
        vis1.count = c->dCounts + 3;
        if (!c->inFrame) CHORD_HIP(c, hipMemsetAsync(c->dCounts + 3, 0, 4, c->stream));
        c->listMine[1] = inMine;
        launch_hzb_cull(c, hb, 1, inL, vis1, nullptr);
        if (outVisible) *outVisible = vis1.handle();
        if (outRejected) *outRejected = ChordCountAndCmd{nullptr, nullptr, 0};
    }
    CHORD_HIP(c, hipGetLastError());
    return CHORDVIS_OK;
}

int chordvis_render_mesh(ChordCtx* c, ChordCountAndCmd in)
{
    int rc = ready(c, "render_mesh");
    if (rc) return rc;
    if (!in.count || !in.cmds) return CHORDVIS_OK;       // {nullptr, nullptr}: nothing to render (instance_culling.cpp:92-95)
    if ((rc = flush_view(c))) return rc;
    return do_raster(c, from_handle(in));
}

int chordvis_visibility_stage0(ChordCtx* c, const ChordHZB* hzbPrev, ChordCountAndCmd in, ChordCountAndCmd* outRejected, int* shouldStage1)
{
    int rc = ready(c, "visibility_stage0");
    if (rc) return rc;
    if (shouldStage1) *shouldStage1 = 0;
    if (outRejected) *outRejected = ChordCountAndCmd{nullptr, nullptr, 0};
    const bool hzbOn = hzbPrev && hzbPrev->minTexels && (c->hView.flags & CHORD_FLAG_HZB_CULL);   // mesh_raster.cpp:293
    if (hzbOn) {
        ChordCountAndCmd vis, rej;
        if ((rc = chordvis_hzb_culling(c, hzbPrev, 1, in, &vis, &rej))) return rc;
        if ((rc = chordvis_render_mesh(c, vis))) return rc;
        if (outRejected) *outRejected = rej;
        if (shouldStage1) *shouldStage1 = 1;
    } else {
        if ((rc = chordvis_render_mesh(c, in))) return rc;
    }
    return CHORDVIS_OK;
}

int chordvis_visibility_stage1(ChordCtx* c, const ChordHZB* hzb, ChordCountAndCmd in)
{
    int rc = ready(c, "visibility_stage1");
    if (rc) return rc;
    ChordCountAndCmd vis;
    if ((rc = chordvis_hzb_culling(c, hzb, 0, in, &vis, nullptr))) return rc;
    return chordvis_render_mesh(c, vis);
}

int chordvis_build_hzb(ChordCtx* c, int bBuildMin, int bBuildMax, int bBuildValidRange, int slot, ChordHZB* out)
{
    int rc = ready(c, "build_hzb");
    if (rc) return rc;
    if (slot < 0 || slot > 2 || !(bBuildMin || bBuildMax) || (bBuildValidRange && !(bBuildMin && bBuildMax)))   // hzb.cpp:43-47
        return fail(c, CHORDVIS_E_INVALID, "build_hzb: slot in 0..2, at least one channel, valid range needs min and max");
    if ((rc = flush_pending_tail(c))) return rc;
    // (a sharded context: from the resolved, row-major image -- chordvis_frame_resolve_visibility / phase c)
    launch_hzb_build(c, c->hzb[slot], bBuildMin != 0, bBuildMax != 0, bBuildValidRange != 0);
    CHORD_HIP(c, hipGetLastError());
    if (out) {
        *out = c->hzb[slot].handle();
        if (!bBuildMax) out->maxTexels = nullptr;
        if (!bBuildValidRange) out->validRange = nullptr;
    }
    return CHORDVIS_OK;
}

int chordvis_reset_history(ChordCtx* c)
{
    if (!c) return CHORDVIS_E_INVALID;
    c->pendingTailSlot = 0;
    c->historySlot = 0;
    return CHORDVIS_OK;
}

static int render_frame_impl(ChordCtx* c)
{
    int rc = ready(c, "render_frame");
    if (rc) return rc;
    if (c->comm) return comm_render_frame(c);            // one process per GPU: the library runs the two all-gathers (RCCL)
    if (c->shard.ranks > 1) return fail(c, CHORDVIS_E_INVALID, "render_frame: a sharded context needs a communicator (chordvis_comm_init_rank), a ChordGroup, or the host drives frame_phase_a/b/c");
    begin_frame_stamps(c);
    if ((rc = begin_frame_clear(c))) return rc;                                       // renderer.cpp:315
    record(c, S_CLEAR);
    ChordHZB hist;
    const bool haveHist = c->historySlot != 0;
    if (haveHist) hist = c->hzb[c->historySlot].handle();
    const int next = c->historySlot == 1 ? 2 : 1;
    ChordCountAndCmd post;
    // (short scenes: instanceCulling and the phase-0 occlusion cull of stage 0 are one kernel -- launch_group_cull is told what stage 0
    // will cull against, and chordvis_hzb_culling finds its lists made)
    c->fuseCullFrame = true;
    c->fuseCullHzb = (haveHist && (c->hView.flags & CHORD_FLAG_HZB_CULL)) ? &c->hzb[c->historySlot] : nullptr;
    c->fusedCullDone = false;
    rc = chordvis_instance_culling(c, &post);                                         // :321 (its first kernel also carries the previous frame's HZB tail)
    c->fuseCullFrame = false; c->fuseCullHzb = nullptr;
    if (rc) return rc;
    record(c, S_CULL);
    // buildHZB is fused into the raster: the tile kernel reduces every finished 64x64 tile to mips 0..5 of the
    // chain kept as history (and of the temporary chain stage 1 culls against); only the one-block tail remains.
    c->fuseHzb = true;
    c->fuseHzbSlot = next;
    c->fuseHzbTemp = haveHist && (c->hView.flags & CHORD_FLAG_HZB_CULL);
    ChordCountAndCmd rejected;
    int stage1 = 0;
    rc = chordvis_visibility_stage0(c, haveHist ? &hist : nullptr, post, &rejected, &stage1);   // :326
    c->fusedCullDone = false;
    record(c, S_STAGE0_END);
    c->shouldStage1 = stage1 != 0;
    if (!rc && stage1) {
        // buildHZB(min) :334 -- levels 0..5 came out of the tile kernel; the one-block tail (levels 6..) is reduced by the
        // blocks of the phase-1 cull themselves unless the chain does not fit their LDS budget (never for targets <= 4096^2)
        const ChordHZBDesc& hd = c->hzb[0].desc;
        uint32_t tailFloats = 0;
        for (uint32_t l = 6; l < hd.mipCount; l++) tailFloats += std::max(1u, hd.width >> l) * std::max(1u, hd.height >> l);
        c->hzbTailInCull = hd.mipCount > 6u && tailFloats <= 1408u && !(c->debugFlags & 131072u);
        if (!c->hzbTailInCull) launch_hzb_tail(c, c->hzb[0], false, false);
        record(c, S_HZB0);
        ChordHZB tmp = c->hzb[0].handle();
        rc = chordvis_visibility_stage1(c, &tmp, rejected);                           // :337
        c->hzbTailInCull = false;
        record(c, S_STAGE1_END);
    }
    c->fuseHzb = false;
    if (rc) return rc;
    // buildHZB(min,max,range) :343 -- mips 0..5 are written; the one-block tail (mips 6.., range) is carried by the next
    // frame's first kernel, or launched by whoever reads the chain first (flush_pending_tail)
    c->hzb[next].valid = true;
    c->pendingTailSlot = next;
    record(c, S_HZBF);
    c->historySlot = next;                                                            // :489
    c->inFrame = false;
    c->lastFrameLaunches = (uint32_t)(c->launchCount - c->frameLaunchBase);
    return CHORDVIS_OK;
}

// Sharded frame.  The single-GPU frame with two differences: the tile kernel writes each tile's words to the
// tile's slot of the rank's chunk and its HZB texels (mips 0..5) to the tile's slots of the two exchange buffers, and after
// each all-gather a copy kernel (launch_hzb_untile) moves every rank's texels to their places in the chain.
//   phase a   clear .. stage 0 raster (owned tiles; fused reduction into the exchange slots)
//   [all-gather of the mid-frame exchange buffer -- only when phase a reported a second stage]
//   phase b   chain 0 from the exchanged texels, phase-1 cull (reduces mips 6.. itself), stage 1 raster
//   [all-gather of the end-of-frame exchange buffer (small) and of the visibility words (the image)]
//   phase c   row-major copy of the image + the history chain from the exchanged texels; ends the frame
// Hosts that let the image travel beside the next frame call chordvis_frame_phase_c_finish once the small exchange has landed
// and chordvis_frame_resolve_visibility whenever the image has.
// The sharded group cull's first half (optional: a frame that starts at phase a culls every group on every rank, as before).
//   phase cull   object pass + the group tests of this rank's range of count blocks -> rank-mask words in its chunk of the cull exchange buffer
//   [all-gather of the cull exchange buffer: chordvis_cull_exchange_ptr, chunk = chordvis_cull_exchange_chunk_bytes]
//   phase a      masks and block counts from the exchanged words, prefix + scatter (the rank's own list), then as above
static int frame_phase_cull_impl(ChordCtx* c)
{
    int rc = ready(c, "frame_phase_cull");
    if (rc) return rc;
    if (c->shard.ranks <= 1) return fail(c, CHORDVIS_E_INVALID, "frame_phase_cull: the context is not sharded");
    if (c->cullPhaseDone) return fail(c, CHORDVIS_E_INVALID, "frame_phase_cull: called twice without chordvis_frame_phase_a in between");
    if (!cull_shardable_config(c)) return fail(c, CHORDVIS_E_INVALID, "frame_phase_cull: the sharded cull needs 2..8 ranks and the flat cull mode (start the frame at chordvis_frame_phase_a instead)");
    if ((rc = ensure_cull_exchange(c))) return rc;       // (made by upload_scene / set_shard already: a no-op)
    begin_frame_stamps(c);
    if ((rc = begin_frame_clear(c))) return rc;
    record(c, S_CLEAR);
    launch_cull_masks(c);
    CHORD_HIP(c, hipGetLastError());
    c->cullPhaseDone = true;
    record(c, S_CULL);
    return CHORDVIS_OK;
}

static int frame_phase_a_impl(ChordCtx* c)
{
    int rc = ready(c, "frame_phase_a");
    if (rc) return rc;
    if (c->shard.ranks <= 1) return fail(c, CHORDVIS_E_INVALID, "frame_phase_a: the context is not sharded");
    if (c->cullPhaseDone) record(c, S_EXCH_CULL);       // time spent in the all-gather of the rank masks (the library's or the caller's)
    else {
        begin_frame_stamps(c);
        if ((rc = begin_frame_clear(c))) return rc;
        record(c, S_CLEAR);
    }
    ChordCountAndCmd post;
    // (the post-cull list is not handed out here: a rank writes only its own share of it; chordvis_last_frame_cmds, the read-back
    // and the tile marker make the full list when they are asked for it -- from the cull's own masks, which every rank holds for every group)
    c->lazyFullList = true;
    rc = chordvis_instance_culling(c, &post);                                        // (its first kernel carries the previous frame's HZB tail)
    c->lazyFullList = false;
    if (rc) return rc;
    record(c, S_CULL);
    ChordHZB hist;
    const bool haveHist = c->historySlot != 0;
    if (haveHist) hist = c->hzb[c->historySlot].handle();
    c->fuseHzb = true;
    c->fuseHzbSlot = c->historySlot == 1 ? 2 : 1;
    c->fuseHzbTemp = haveHist && (c->hView.flags & CHORD_FLAG_HZB_CULL);
    c->exchangeSlotsFresh = true;                        // (the fused tile kernel of this frame's passes writes the rank's exchange slots)
    ChordCountAndCmd rejected;
    int stage1 = 0;
    rc = chordvis_visibility_stage0(c, haveHist ? &hist : nullptr, post, &rejected, &stage1);
    c->fuseHzb = false;
    if (rc) return rc;
    c->shouldStage1 = stage1 != 0;
    c->lastRejected = from_handle(rejected);
    record(c, S_STAGE0_END);
    return CHORDVIS_OK;
}

// phase b: [mid-frame exchange buffer all-gathered by the caller] -> HZB chain 0 -> stage 1.
static int frame_phase_b_impl(ChordCtx* c)
{
    int rc = ready(c, "frame_phase_b");
    if (rc) return rc;
    if (!c->shouldStage1) return CHORDVIS_OK;
    record(c, S_EXCH_HZB);       // time spent in the all-gather of the mid-frame exchange buffer (the library's or the caller's)
    launch_hzb_untile(c, c->hzb[0], false);
    CHORD_HIP(c, hipGetLastError());
    {   // levels 6.. of the chain: reduced by the blocks of the phase-1 cull themselves when they fit (render_frame_impl)
        const ChordHZBDesc& hd = c->hzb[0].desc;
        uint32_t tailFloats = 0;
        for (uint32_t l = 6; l < hd.mipCount; l++) tailFloats += std::max(1u, hd.width >> l) * std::max(1u, hd.height >> l);
        c->hzbTailInCull = hd.mipCount > 6u && tailFloats <= 1408u && !(c->debugFlags & 131072u);
        if (!c->hzbTailInCull) launch_hzb_tail(c, c->hzb[0], false, false);
    }
    record(c, S_HZB0);
    ChordHZB tmp = c->hzb[0].handle();
    c->fuseHzb = true;
    c->fuseHzbTemp = false;
    rc = chordvis_visibility_stage1(c, &tmp, c->lastRejected.handle());
    c->fuseHzb = false;
    c->hzbTailInCull = false;
    if (rc) return rc;
    record(c, S_STAGE1_END);
    return CHORDVIS_OK;
}

// [end-of-frame exchange buffer all-gathered by the caller] -> the history chain; the frame is over.  Mips 0..5 are copied
// from the slots; the one-block tail (mips 6.., valid range from the tiles' pairs) rides on the next frame's first kernel, or
// is launched by whoever reads the chain first (flush_pending_tail) -- as in the single-GPU frame.
static int frame_phase_c_finish_impl(ChordCtx* c)
{
    int rc = ready(c, "frame_phase_c_finish");
    if (rc) return rc;
    if (c->shard.ranks <= 1) return fail(c, CHORDVIS_E_INVALID, "frame_phase_c_finish: the context is not sharded");
    // the history chain, the tiles' valid ranges and their loads come from the end-of-frame exchange slots, which only the fused
    // tile kernel of chordvis_frame_phase_a / _b writes: a frame rastered with the stand-alone passes has none to unpack
    if (!c->exchangeSlotsFresh) return fail(c, CHORDVIS_E_INVALID, "frame_phase_c: no chordvis_frame_phase_a in this frame -- the end-of-frame exchange slots are stale (a frame rastered with the stand-alone passes builds its history with chordvis_build_hzb from the resolved image)");
    c->exchangeSlotsFresh = false;
    const int next = c->historySlot == 1 ? 2 : 1;
    launch_hzb_untile(c, c->hzb[next], true);
    CHORD_HIP(c, hipGetLastError());
    c->pendingTailSlot = next;
    record(c, S_HZBF);
    c->historySlot = next;
    c->inFrame = false;
    c->lastFrameLaunches = (uint32_t)(c->launchCount - c->frameLaunchBase);
    return CHORDVIS_OK;
}

// phase c: [both end-of-frame all-gathers done by the caller] -> row-major copy of the image + the history chain.
static int frame_phase_c_impl(ChordCtx* c)
{
    int rc = ready(c, "frame_phase_c");
    if (rc) return rc;
    if (c->shard.ranks <= 1) return fail(c, CHORDVIS_E_INVALID, "frame_phase_c: the context is not sharded");
    record(c, S_EXCH_VIS);       // time spent in the end-of-frame all-gathers
    launch_detile(c);
    CHORD_HIP(c, hipGetLastError());
    return frame_phase_c_finish_impl(c);
}

// A failed frame must not leave frame-scoped state behind (the stand-alone passes that may follow would skip their
// count resets, a later frame would inherit a fused-HZB request).
static int end_failed_frame(ChordCtx* c, int rc)
{
    if (rc && c) {
        if (c->zeroFrameStateInCull) { (void)hipMemsetAsync(c->dFrameState, 0, c->frameStateZeroBytes, c->stream); c->zeroFrameStateInCull = false; }
        c->inFrame = false; c->fuseHzb = false; c->pendingClear = false; c->shouldStage1 = false; c->cullPhaseDone = false; c->exchangeSlotsFresh = false;
    }
    return rc;
}
int chordvis_render_frame(ChordCtx* c) { return end_failed_frame(c, render_frame_impl(c)); }
int chordvis_frame_phase_cull(ChordCtx* c) { return end_failed_frame(c, frame_phase_cull_impl(c)); }
int chordvis_frame_phase_a(ChordCtx* c) { return end_failed_frame(c, frame_phase_a_impl(c)); }
int chordvis_frame_phase_b(ChordCtx* c) { return end_failed_frame(c, frame_phase_b_impl(c)); }
int chordvis_frame_phase_c(ChordCtx* c) { return end_failed_frame(c, frame_phase_c_impl(c)); }
int chordvis_frame_phase_c_finish(ChordCtx* c) { return end_failed_frame(c, frame_phase_c_finish_impl(c)); }

// The row-major copy of the (gathered) rank-major visibility words of the CURRENT buffer pair, on `hipStream` (NULL: the
// context's stream) -- phase c's first half, for pipelined hosts that run it beside the next frame.
int chordvis_frame_resolve_visibility(ChordCtx* c, void* hipStream)
{
    if (!c || !c->dVis || c->shard.ranks <= 1) return fail(c, CHORDVIS_E_INVALID, "frame_resolve_visibility: a sharded context with a gbuffer");
    launch_detile(c, (hipStream_t)hipStream);
    CHORD_HIP(c, hipGetLastError());
    return CHORDVIS_OK;
}

// Orders `hipStream` (NULL: the context's stream) behind the completion of the resolved image of the last submitted frame.
// Needed only after pipelined ChordGroup frames, whose image is gathered beside the context's stream; a no-op otherwise.
int chordvis_wait_visibility(ChordCtx* c, void* hipStream)
{
    if (!c || !c->dVis) return fail(c, CHORDVIS_E_INVALID, "wait_visibility: no gbuffer");
    hipStream_t s = hipStream ? (hipStream_t)hipStream : c->stream;
    if (c->visReadyEvent[0]) CHORD_HIP(c, hipStreamWaitEvent(s, c->visReadyEvent[0], 0));
    else if (s != c->stream) {
        // not pipelined: the image is complete when the context's stream is; order the foreign stream behind it
        hipEvent_t e = nullptr;
        CHORD_HIP(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        hipError_t rc = hipEventRecord(e, c->stream);
        if (rc == hipSuccess) rc = hipStreamWaitEvent(s, e, 0);
        (void)hipEventDestroy(e);                                    // (destruction is deferred until the event has completed)
        if (rc != hipSuccess) return fail(c, CHORDVIS_E_HIP, "wait_visibility", rc);
    }
    return CHORDVIS_OK;
}

int chordvis_last_frame_cmds(ChordCtx* c, ChordCountAndCmd* out)
{
    if (!c || !out || !c->sceneLoaded) return fail(c, CHORDVIS_E_INVALID, "last_frame_cmds: no scene");
    launch_full_list(c);                                             // (a sharded frame wrote only the rank's share of it)
    CHORD_HIP(c, hipGetLastError());
    *out = c->lists[0].handle();
    return CHORDVIS_OK;
}

int chordvis_history_hzb(ChordCtx* c, ChordHZB* out)
{
    if (!c || !out || c->historySlot == 0) return fail(c, CHORDVIS_E_INVALID, "history_hzb: no history yet");
    { const int rc = flush_pending_tail(c); if (rc) return rc; }
    *out = c->hzb[c->historySlot].handle();
    return CHORDVIS_OK;
}

// ------------------------------------------------------------------------ tile marker (8f-1) --

int chordvis_visibility_mark(ChordCtx* c, ChordCountAndCmd drawed, ChordTileMarker* out)
{
    if (!c || !out || !c->dVis || !c->sceneLoaded) return fail(c, CHORDVIS_E_INVALID, "visibility_mark: no scene / gbuffer");
    if (!drawed.count || !drawed.cmds) return fail(c, CHORDVIS_E_INVALID, "visibility_mark: null command list");
    if (drawed.cmds == c->lists[0].cmds) { launch_full_list(c); CHORD_HIP(c, hipGetLastError()); }
    const uint32_t mW = (c->width + 7u) / 8u, mH = (c->height + 7u) / 8u;
    int rc;
    if (!c->dTileMarker) {
        if ((rc = dalloc(c, &c->dTileMarker, (size_t)mW * mH * 4))) return rc;
        if ((rc = dalloc(c, &c->dShadingTiles, (size_t)mW * mH * 2 + 8))) return rc;
    }
    const unsigned long long* vis = (const unsigned long long*)(c->shard.ranks > 1 ? c->dVisResolved : c->dVis);
    // (pipelined group frames: the image is gathered and resolved beside the context's stream)
    if (c->visReadyEvent[0]) CHORD_HIP(c, hipStreamWaitEvent(c->stream, c->visReadyEvent[0], 0));
    chord::launch_visibility_mark(c, vis, drawed.cmds, drawed.count, c->dTileMarker);
    CHORD_HIP(c, hipGetLastError());
    out->marker = c->dTileMarker;
    out->visibilityDim[0] = c->width; out->visibilityDim[1] = c->height;
    out->markerDim[0] = mW; out->markerDim[1] = mH;
    return CHORDVIS_OK;
}

int chordvis_prepare_shading_tile_param(ChordCtx* c, uint32_t shadingType, const ChordTileMarker* marker, ChordShadingTiles* out)
{
    if (!c || !marker || !out || !marker->marker || marker->marker != c->dTileMarker)
        return fail(c, CHORDVIS_E_INVALID, "prepare_shading_tile_param: marker is not this context's");
    if (shadingType >= 128u) return fail(c, CHORDVIS_E_INVALID, "prepare_shading_tile_param: shading type does not fit the 128-bit marker");
    const uint32_t total = marker->markerDim[0] * marker->markerDim[1];
    uint32_t* count = c->dShadingTiles + (size_t)total * 2;
    chord::launch_shading_tiles(c, c->dTileMarker, shadingType, c->dShadingTiles, count, count + 4);
    CHORD_HIP(c, hipGetLastError());
    out->tileCmd = c->dShadingTiles; out->count = count; out->dispatchIndirect = count + 4; out->capacity = total;
    return CHORDVIS_OK;
}

int chordvis_readback_tile_marker(ChordCtx* c, const ChordTileMarker* marker, uint32_t* host)
{
    if (!c || !marker || !host || !marker->marker) return fail(c, CHORDVIS_E_INVALID, "readback_tile_marker: null argument");
    CHORD_HIP(c, hipStreamSynchronize(c->stream));
    CHORD_HIP(c, hipMemcpy(host, marker->marker, (size_t)marker->markerDim[0] * marker->markerDim[1] * 16, hipMemcpyDeviceToHost));
    return CHORDVIS_OK;
}

int chordvis_readback_shading_tiles(ChordCtx* c, const ChordShadingTiles* tiles, uint32_t* hostTiles, uint32_t hostCapacity,
                                    uint32_t* hostCount, uint32_t hostArgs[4])
{
    if (!c || !tiles || !tiles->count || !hostCount) return fail(c, CHORDVIS_E_INVALID, "readback_shading_tiles: null argument");
    CHORD_HIP(c, hipStreamSynchronize(c->stream));
    CHORD_HIP(c, hipMemcpy(hostCount, tiles->count, sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (hostArgs) CHORD_HIP(c, hipMemcpy(hostArgs, tiles->dispatchIndirect, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    const uint32_t n = *hostCount < hostCapacity ? *hostCount : hostCapacity;
    if (hostTiles && n) CHORD_HIP(c, hipMemcpy(hostTiles, tiles->tileCmd, (size_t)n * 8, hipMemcpyDeviceToHost));
    return CHORDVIS_OK;
}

// ------------------------------------------------------------------------------ readback/stats --

int chordvis_readback_visibility(ChordCtx* c, uint64_t* host)
{
    if (!c || !host || !c->dVis) return fail(c, CHORDVIS_E_INVALID, "readback_visibility: no gbuffer");
    CHORD_HIP(c, hipStreamSynchronize(c->stream));
    if (c->visReadyEvent[0]) CHORD_HIP(c, hipEventSynchronize(c->visReadyEvent[0]));   // (pipelined group: the image is resolved beside the stream)
    const uint64_t* src = c->shard.ranks > 1 ? c->dVisResolved : c->dVis;
    CHORD_HIP(c, hipMemcpy(host, src, (size_t)c->width * c->height * 8, hipMemcpyDeviceToHost));
    return CHORDVIS_OK;
}

// Pipelined sharded frames: the image of the frame BEFORE the last submitted one (the other buffer pair).
int chordvis_readback_previous_visibility(ChordCtx* c, uint64_t* host)
{
    if (!c || !host || !c->dVisResolvedAlt) return fail(c, CHORDVIS_E_INVALID, "readback_previous_visibility: no second frame in flight (chordvis_swap_visibility)");
    CHORD_HIP(c, hipStreamSynchronize(c->stream));
    if (c->visReadyEvent[1]) CHORD_HIP(c, hipEventSynchronize(c->visReadyEvent[1]));
    CHORD_HIP(c, hipMemcpy(host, c->dVisResolvedAlt, (size_t)c->width * c->height * 8, hipMemcpyDeviceToHost));
    return CHORDVIS_OK;
}

int chordvis_readback_cmds(ChordCtx* c, ChordCountAndCmd h, ChordDrawCmd* host, uint32_t cap, uint32_t* outCount)
{
    if (!c || !h.count || !h.cmds || !outCount) return fail(c, CHORDVIS_E_INVALID, "readback_cmds: invalid handle");
    if (h.cmds == c->lists[0].cmds) { launch_full_list(c); CHORD_HIP(c, hipGetLastError()); }
    CHORD_HIP(c, hipStreamSynchronize(c->stream));
    uint32_t n = 0;
    CHORD_HIP(c, hipMemcpy(&n, h.count, 4, hipMemcpyDeviceToHost));
    *outCount = n;
    const uint32_t m = std::min(std::min(n, cap), h.capacity);
    if (host && m) {
        CHORD_HIP(c, hipMemcpy(host, h.cmds, sizeof(ChordDrawCmd) * m, hipMemcpyDeviceToHost));
        // device meshlet ids are flattened over assets; hand back asset-relative ids (the reference's cmd.y)
        for (uint32_t i = 0; i < m; i++) {
            if (host[i].objectId < c->objectCount) host[i].meshletId -= c->hPrims[c->hObjStatic[host[i].objectId].prim].assetMeshletBase;
        }
    }
    return CHORDVIS_OK;
}

int chordvis_readback_hzb(ChordCtx* c, const ChordHZB* hzb, uint16_t* hostMin, uint16_t* hostMax, uint32_t hostValidRange[2])
{
    if (!c || !hzb) return fail(c, CHORDVIS_E_INVALID, "readback_hzb: null handle");
    { const int rc = flush_pending_tail(c); if (rc) return rc; }
    CHORD_HIP(c, hipStreamSynchronize(c->stream));
    const size_t bytes = sizeof(uint16_t) * hzb->desc.totalTexels;
    if (hostMin && hzb->minTexels) CHORD_HIP(c, hipMemcpy(hostMin, hzb->minTexels, bytes, hipMemcpyDeviceToHost));
    if (hostMax && hzb->maxTexels) CHORD_HIP(c, hipMemcpy(hostMax, hzb->maxTexels, bytes, hipMemcpyDeviceToHost));
    if (hostValidRange && hzb->validRange) CHORD_HIP(c, hipMemcpy(hostValidRange, hzb->validRange, 8, hipMemcpyDeviceToHost));
    return CHORDVIS_OK;
}

int chordvis_upload_history_hzb(ChordCtx* c, const uint16_t* hostMin)
{
    if (!c || !hostMin || !c->dVis) return fail(c, CHORDVIS_E_INVALID, "upload_history_hzb: no gbuffer");
    c->pendingTailSlot = 0;                              // the uploaded chain replaces whatever was pending
    const int slot = c->historySlot == 1 ? 2 : 1;
    CHORD_HIP(c, hipStreamSynchronize(c->stream));
    CHORD_HIP(c, hipMemcpy(c->hzb[slot].minTexels, hostMin, sizeof(uint16_t) * c->hzb[slot].desc.totalTexels, hipMemcpyHostToDevice));
    c->hzb[slot].valid = true;
    c->historySlot = slot;
    return CHORDVIS_OK;
}

int chordvis_set_debug(ChordCtx* c, uint32_t flags)
{
    if (c) c->orderAge = c->orderAge1 = 0xFFFFFFFFu;                    // (the kept tile schedule of launch_raster is made again)

    if (!c) return CHORDVIS_E_INVALID;
    // measurement switches the library was not built with would silently measure the product: refuse them
    if (!RASTER_PROFILE && (flags & CHORD_DEBUG_PROFILE_BITS))
        return fail(c, CHORDVIS_E_INVALID, "set_debug: the phase clocks (bits 16, 512) need a library built with -DRASTER_PROFILE=1 (chord_amd/build.py --tag prof -DRASTER_PROFILE=1)");
    if (!RASTER_ABLATION && (flags & CHORD_DEBUG_ABLATION_BITS))
        return fail(c, CHORDVIS_E_INVALID, "set_debug: the ablation switches need a library built with -DRASTER_ABLATION=1 (chord_amd/build.py --tag abl -DRASTER_ABLATION=1)");
    c->debugFlags = flags;
    if (c->depthCtx) c->depthCtx->debugFlags = flags;      // (the depth views' child context follows)
    return CHORDVIS_OK;
}

int chordvis_debug_tile_profile(ChordCtx* c, int pass, uint64_t* hostTicks, uint32_t* hostCounts, uint32_t capacity)
{
    if (!c || pass < 0 || pass > 1 || !hostTicks || !hostCounts || capacity < c->tilesX * c->tilesY) return fail(c, CHORDVIS_E_INVALID, "debug_tile_profile: bad arguments");
    CHORD_HIP(c, hipStreamSynchronize(c->stream));
    const size_t n = (size_t)c->tilesX * c->tilesY;
    CHORD_HIP(c, hipMemcpy(hostTicks, c->dTileClocks + (size_t)pass * CHORD_MAX_TILES, n * 8, hipMemcpyDeviceToHost));
    CHORD_HIP(c, hipMemcpy2D(hostCounts, 4, c->dFrameState->tileCount + (size_t)pass * c->tilesX * c->tilesY * CHORD_TILECOUNT_STRIDE, 4 * CHORD_TILECOUNT_STRIDE, 4, n, hipMemcpyDeviceToHost));
    if (capacity >= 9 * n)   // optional: 8 phase accumulators per tile follow the totals
        CHORD_HIP(c, hipMemcpy(hostTicks + n, c->dTileClocks + (size_t)2 * CHORD_MAX_TILES + (size_t)pass * CHORD_MAX_TILES * 8, n * 64, hipMemcpyDeviceToHost));
    return CHORDVIS_OK;
}

// Measurement aid: captures two consecutive frames (the history slot alternates) into a hipGraph and replays it.
int chordvis_debug_graph_frames(ChordCtx* c, uint32_t pairs, float* msPerFrameStream, float* msPerFrameGraph)
{
    if (!c || !pairs || !msPerFrameStream || !msPerFrameGraph) return fail(c, CHORDVIS_E_INVALID, "debug_graph_frames: bad arguments");
    if (!c->ownStream) return fail(c, CHORDVIS_E_INVALID, "debug_graph_frames: needs a context-owned (capturable) stream");
    int rc;
    // (the kept tile schedule is off for both measurements: a captured frame either holds the schedule kernel or it does not, whatever the
    // age of the schedule at replay -- with it in every frame the stream's and the graph's frames are the same eleven + one launches)
    struct KeepOff { ChordCtx* c; uint32_t keep; ~KeepOff() { c->orderKeepFrames = keep; c->orderAge = c->orderAge1 = 0xFFFFFFFFu; } } keepOff{c, c->orderKeepFrames};
    c->orderKeepFrames = 0u;
    for (int i = 0; i < 4; i++) if ((rc = chordvis_render_frame(c))) return rc;
    hipEvent_t e0, e1;
    CHORD_HIP(c, hipEventCreate(&e0)); CHORD_HIP(c, hipEventCreate(&e1));
    CHORD_HIP(c, hipStreamSynchronize(c->stream));
    CHORD_HIP(c, hipEventRecord(e0, c->stream));
    for (uint32_t i = 0; i < 2 * pairs; i++) if ((rc = chordvis_render_frame(c))) return rc;
    CHORD_HIP(c, hipEventRecord(e1, c->stream));
    CHORD_HIP(c, hipEventSynchronize(e1));
    float ms = 0;
    CHORD_HIP(c, hipEventElapsedTime(&ms, e0, e1));
    *msPerFrameStream = ms / (2.0f * pairs);
    hipGraph_t graph; hipGraphExec_t exec;
    CHORD_HIP(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed));
    rc = chordvis_render_frame(c);
    if (!rc) rc = chordvis_render_frame(c);
    hipError_t ce = hipStreamEndCapture(c->stream, &graph);
    if (rc) return rc;
    CHORD_HIP(c, ce);
    CHORD_HIP(c, hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    for (int i = 0; i < 4; i++) CHORD_HIP(c, hipGraphLaunch(exec, c->stream));
    CHORD_HIP(c, hipStreamSynchronize(c->stream));
    CHORD_HIP(c, hipEventRecord(e0, c->stream));
    for (uint32_t i = 0; i < pairs; i++) CHORD_HIP(c, hipGraphLaunch(exec, c->stream));
    CHORD_HIP(c, hipEventRecord(e1, c->stream));
    CHORD_HIP(c, hipEventSynchronize(e1));
    CHORD_HIP(c, hipEventElapsedTime(&ms, e0, e1));
    *msPerFrameGraph = ms / (2.0f * pairs);
    (void)hipGraphExecDestroy(exec); (void)hipGraphDestroy(graph); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return CHORDVIS_OK;
}

// Debugging aid: raw read of an internal buffer.  which: 0 tile counts (FrameState::tileCount), 1 fixed bins, 2 chunk table,
// 3 bin pool, 4 compact records, 5 wide records, 6 the fused cull kernel's look-back words.  offset / bytes in bytes.
int chordvis_debug_read(ChordCtx* c, int which, uint64_t offset, uint64_t bytes, void* host)
{
    if (!c || !host) return fail(c, CHORDVIS_E_INVALID, "debug_read: null argument");
    CHORD_HIP(c, hipStreamSynchronize(c->stream));
    const char* base = nullptr;
    switch (which) {
    case 0: base = (const char*)c->dFrameState->tileCount; break;
    case 1: base = (const char*)c->dTileBins; break;
    case 2: base = (const char*)c->dBinChunkTab; break;
    case 3: base = (const char*)c->dBinPool; break;
    case 4: base = (const char*)c->dTrisC; break;
    case 5: base = (const char*)c->dTris; break;
    case 6: base = (const char*)c->dCullLookback; break;       // look-back words of frame_cull_fused_kernel (profile build: stage clocks behind them)
    default: return fail(c, CHORDVIS_E_INVALID, "debug_read: unknown buffer");
    }
    CHORD_HIP(c, hipMemcpy(host, base + offset, bytes, hipMemcpyDeviceToHost));
    return CHORDVIS_OK;
}

// Measurement / debugging aid: words of the split-tile accumulation slabs that are not zero (the invariant between
// raster passes is: none).
int chordvis_debug_slab_nonzero(ChordCtx* c, uint64_t* count)
{
    if (!c || !count || !c->dTileSlabs) return fail(c, CHORDVIS_E_INVALID, "debug_slab_nonzero: no gbuffer");
    CHORD_HIP(c, hipStreamSynchronize(c->stream));
    const size_t n = (size_t)c->tilesX * c->tilesY * CHORD_TILE * CHORD_TILE;
    std::vector<unsigned long long> h(n);
    CHORD_HIP(c, hipMemcpy(h.data(), c->dTileSlabs, n * 8, hipMemcpyDeviceToHost));
    uint64_t k = 0;
    for (size_t i = 0; i < n; i++) k += h[i] != 0ull;
    *count = k;
    return CHORDVIS_OK;
}

int chordvis_debug_setup_profile(ChordCtx* c, int pass, uint64_t hostTicks[5], uint32_t* waves)
{
    if (!c || pass < 0 || pass > 1 || !hostTicks) return fail(c, CHORDVIS_E_INVALID, "debug_setup_profile: bad arguments");
    CHORD_HIP(c, hipStreamSynchronize(c->stream));
    const size_t n = (size_t)CHORD_MAX_TILES * 8;
    std::vector<uint64_t> v(n);
    CHORD_HIP(c, hipMemcpy(v.data(), c->dTileClocks + (size_t)2 * CHORD_MAX_TILES + (size_t)pass * CHORD_MAX_TILES * 8, n * 8, hipMemcpyDeviceToHost));
    const uint32_t w = (uint32_t)std::min<size_t>((size_t)c->numCUs * 6 * 4, n / 5);
    for (int i = 0; i < 5; i++) hostTicks[i] = 0;
    for (uint32_t k = 0; k < w; k++) for (int i = 0; i < 5; i++) hostTicks[i] += v[(size_t)k * 5 + i];
    if (waves) *waves = w;
    return CHORDVIS_OK;
}

int chordvis_enable_timers(ChordCtx* c, int enable)
{
    if (!c) return CHORDVIS_E_INVALID;
    c->timers = enable < 0 ? 0 : ((enable & 0xFF) > 2 ? 2 : (enable & 0xFF));
    c->timerPeriod = (enable >> 8) > 0 ? (uint32_t)(enable >> 8) : 1u;
    c->stampTags.clear();
    c->framesStamped = 0;
    c->frameIndex = 0;
    c->stampThisFrame = false;
    return CHORDVIS_OK;
}

int chordvis_stats(ChordCtx* c, ChordStats* out)
{
    if (!c || !out) return fail(c, CHORDVIS_E_INVALID, "stats: null argument");
    std::memset(out, 0, sizeof(*out));
    { const int rc = flush_pending_tail(c); if (rc) return rc; }
    CHORD_HIP(c, hipStreamSynchronize(c->stream));
    uint32_t counts[4] = {0, 0, 0, 0};
    CHORD_HIP(c, hipMemcpy(counts, c->dCounts, sizeof(counts), hipMemcpyDeviceToHost));
    DeviceCounters dc;
    CHORD_HIP(c, hipMemcpy(&dc, c->dCounters, sizeof(dc), hipMemcpyDeviceToHost));
    out->countInstanceCulled = counts[0];
    if (c->shouldStage1) {
        out->countStage0Visible = counts[1]; out->countStage0Rejected = counts[2]; out->countStage1Visible = counts[3];
        out->trianglesSubmitted = dc.trisHzbVisible0 + dc.trisHzbVisible1;
    } else {
        out->countStage0Visible = counts[0];
        out->trianglesSubmitted = dc.trisInstanceCulled;
    }
    out->overflow = dc.overflow;
    out->rasterLaunches = c->rasterCalls;
    out->kernelLaunches = c->lastFrameLaunches;
    for (int pass = 0; pass < 2; pass++) {
        out->clipTriangles[pass] = dc.clipTriCount[pass];
        for (uint32_t i = 0; i < CHORD_LIST_SHARDS; i++) out->largeRecords[pass] += dc.largeCount[pass][i * CHORD_SHARD_STRIDE];
    }
    for (uint32_t i = 0; i < CHORD_LIST_SHARDS; i++) {
        out->triangleRecords += dc.triCount[i * CHORD_SHARD_STRIDE] + dc.triCountC[i * CHORD_SHARD_STRIDE];
        out->triangleRecordsCompact += dc.triCountC[i * CHORD_SHARD_STRIDE];
        out->pixelBlockBytes += (uint64_t)dc.blockGranules[i * CHORD_SHARD_STRIDE] * 16u;
    }
    if (c->tilesX) {
        std::vector<uint32_t> tc((size_t)c->tilesX * c->tilesY);
        for (int pass = 0; pass < 2; pass++) {
            CHORD_HIP(c, hipMemcpy2D(tc.data(), 4, c->dFrameState->tileCount + (size_t)pass * c->tilesX * c->tilesY * CHORD_TILECOUNT_STRIDE, 4 * CHORD_TILECOUNT_STRIDE, 4, tc.size(), hipMemcpyDeviceToHost));
            for (uint32_t v : tc) { out->binEntries += v; out->tilesTouched[pass] += v ? 1u : 0u; }
            CHORD_HIP(c, hipMemcpy2D(tc.data(), 4, c->dFrameState->tileCount + (size_t)pass * c->tilesX * c->tilesY * CHORD_TILECOUNT_STRIDE + 1, 4 * CHORD_TILECOUNT_STRIDE, 4, tc.size(), hipMemcpyDeviceToHost));
            for (uint32_t v : tc) out->pixelBlocks += v;
        }
    }
    if (c->timers && c->framesStamped && c->stampTags.size() > 1) {
        // walk the stamps: the segment ending at stamp i is attributed by its tag and the current stage
        float clear = 0, cull = 0, st0 = 0, hzb0 = 0, st1 = 0, hzbf = 0, rc_ = 0, rk = 0, rh = 0, other = 0, exh = 0, exv = 0, exc = 0, exf = 0;
        int stage = 0;   // 0 = stage 0, 1 = stage 1
        for (size_t i = 1; i < c->stampTags.size(); i++) {
            const int tag = c->stampTags[i];
            if (tag == S_FRAME_BEGIN) { stage = 0; continue; }           // gap between frames is not frame time
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, c->evPool[i - 1], c->evPool[i]) != hipSuccess) continue;
            switch (tag) {
            case S_CLEAR: clear += ms; break;
            case S_CULL: cull += ms; break;
            case S_HZBCULL: case S_STAGE0_END: case S_STAGE1_END: (stage == 0 ? st0 : st1) += ms; break;
            case S_R_CLUSTER: rc_ += ms; (stage == 0 ? st0 : st1) += ms; break;
            case S_R_CLIP: rk += ms; (stage == 0 ? st0 : st1) += ms; break;
            case S_R_CHUNK: rh += ms; (stage == 0 ? st0 : st1) += ms; break;
            case S_HZB0: hzb0 += ms; break;
            case S_HZBF: hzbf += ms; break;
            case S_EXCH_HZB: exh += ms; break;
            case S_EXCH_VIS: exv += ms; break;
            case S_EXCH_CULL: exc += ms; break;
            case S_EXCH_FINAL: exf += ms; break;
            default: other += ms; break;
            }
            if (tag == S_STAGE0_END) stage = 1;
        }
        const float inv = 1.0f / (float)c->framesStamped;
        out->msClear = clear * inv; out->msInstanceCulling = cull * inv; out->msStage0 = st0 * inv;
        out->msHzbStage0 = hzb0 * inv; out->msStage1 = st1 * inv; out->msHzbFinal = hzbf * inv;
        out->msRasterCluster = rc_ * inv; out->msRasterClip = rk * inv; out->msRasterChunk = rh * inv;
        out->msExchangeHzb = exh * inv; out->msExchangeVis = exv * inv; out->msExchangeCull = exc * inv; out->msExchangeFinal = exf * inv;
        out->msFrame = (clear + cull + st0 + hzb0 + st1 + hzbf + other + exh + exv + exc + exf) * inv;
        out->framesTimed = c->framesStamped;
        out->stampsPerFrame = (float)c->stampTags.size() * inv;
    }
    if (c->timers == 2) { c->stampTags.clear(); c->framesStamped = 0; }
    if (dc.overflow) return fail(c, CHORDVIS_E_CAPACITY, "a deferred raster list overflowed this frame; the visibility buffer is incomplete");
    return CHORDVIS_OK;
}

} // extern "C"
