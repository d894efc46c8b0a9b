// Thin HIP device layer under the pass surface: the counterpart of the reference's
// Vulkan layer in source/graphics (Context graphics.h:88-345, buffer pool
// buffer_pool.h:16-144, queue command_list.h:69-199, GPU timestamps query.cpp:5-127).
//
// One context = one device + one stream.  All buffers are plain hipMalloc
// allocations sized from scene upper bounds at upload time (the reference sizes
// its command buffers from lod0MeshletCount the same way, instance_culling.cpp:141).
#pragma once

#include <hip/hip_runtime.h>
#include <cstddef>

#include <string>
#include <vector>

#include "chordvis.h"

// Build switches of the measurement code in the raster kernels (kernels_raster.hip): per-phase clocks (chordvis_set_debug
// bits 16 and 512) and ablation switches (bits 1, 2, 32, 64, 128, 256, 1024, 2048, 4096, 8192, 16384).  Off in the product
// library; python chord_amd/build.py --tag NAME -DRASTER_PROFILE=1 / -DRASTER_ABLATION=1 builds a measuring one.
#ifndef RASTER_PROFILE
#define RASTER_PROFILE 0
#endif
#ifndef RASTER_ABLATION
#define RASTER_ABLATION 0
#endif
#define CHORD_DEBUG_PROFILE_BITS (16u | 512u)
#define CHORD_DEBUG_ABLATION_BITS (1u | 2u | 32u | 64u | 128u | 256u | 1024u | 2048u | 4096u | 8192u | 16384u | 1048576u | 2097152u)

namespace chord {

// ---- flattened, pre-resolved scene records (bindless indirections removed at upload) ------

struct DPrim {                 // per GLTFPrimitiveBuffer, 48 B
    float    posMin[3]; uint32_t meshletBase;       // asset meshlet base + primitive.meshletOffset
    float    posMax[3]; uint32_t groupBase;         // asset group base + primitive.meshletGroupOffset
    uint32_t groupIndicesBase;                      // asset index base + primitive.meshletGroupIndicesOffset
    uint32_t groupCount;
    uint32_t assetMeshletBase;                      // to convert device meshlet ids back to asset-relative ones
    uint32_t bvhBase;                               // first node of the primitive's tree in dBvhNodes, 0xFFFFFFFF: none
};

struct DBVHNode {              // GPUBVHNode (60 B) padded to one 64-byte line
    float    sphere[4];
    uint32_t children[8];
    uint32_t bvhNodeCount, leafGroupOffset, leafGroupCount, pad;
};

struct DGroup {                // GPUGLTFMeshletGroup padded to 48 B for 16-B loads
    float    clusterPosCenter[3]; float parentError;
    float    parentPosCenter[3];  float error;
    uint32_t meshletOffset; uint32_t meshletCount; uint32_t pad0; uint32_t pad1;
};

struct DMeshlet {              // GPUGLTFMeshlet with dataOffset globalised and lod -> vertexBase
    float    posMin[3]; uint32_t dataOffset;
    float    posMax[3]; uint32_t vertexTriangleCount;
    float    coneAxis[3]; float coneCutOff;
    float    coneApex[3]; uint32_t vertexBase;      // asset vertex base + primitive.vertexOffset
};

struct DObjStatic {            // per object, derived once per upload, 48 B
    uint32_t prim;
    uint32_t matFlags;         // bit 0 bTwoSided, bits 1-2 alphaMode (0 opaque, 1 mask, 2 blend), bits 8.. material index
    uint32_t groupBase;        // first flattened (object, group) index
    uint32_t shadingType;      // materials[object.GLTFMaterialData].materialType (visibility_tile.hlsl:56-60)
    float    posMin[3]; uint32_t pad0;   // the primitive's bounds (DPrim::posMin / posMax) beside the object: one dependent fetch less for
    float    posMax[3]; uint32_t pad1;   // whoever tests the object's box with only the object's index in hand (frame_cull_fused_kernel)
};
static_assert(sizeof(DObjStatic) == 48, "DObjStatic");

#define CHORD_MATFLAG_TWO_SIDED 1u
#define CHORD_MATFLAG_ALPHA(f) (((f) >> 1) & 3u)
#define CHORD_MATFLAG_MATERIAL(f) ((f) >> 8)

// What the masked buckets read of a material (mesh_raster.hlsl:107-112,198-204), texture and sampler pre-resolved.
// One level of a material's alpha texture as the row units of the tile kernel want it (resolved at upload, copied into a masked
// triangle's extension record by the setup kernel).  A non-power-of-two wrap is a remainder by a divisor known here: magic =
// floor(2^32 / period) (period = the size, or twice the size for MIRRORED_REPEAT) turns it into one multiply-high and one
// correction, bias = a multiple of the period >= 2^30 makes the texel index (|i| <= 1e9 + 1) non-negative first.
struct DMatLevel {             // 24 B
    uint32_t base;             // first alpha byte of the level in dTexAlpha
    uint32_t dims;             // (width - 1) | (height - 1) << 16
    uint32_t magicS, biasS, magicT, biasT;   // 0 for a power-of-two period (wrapped with a mask) and for CLAMP_TO_EDGE
};
#define CHORD_MAX_TEX_LEVELS 15u
struct DMaterial {             // 48 + 15 x 24 B
    uint32_t texOffset;        // first alpha byte of level 0 in dTexAlpha; 0xFFFFFFFF: white fallback (alpha 1)
    uint32_t texWidth, texHeight, texMips;
    uint32_t minFilter, magFilter, wrapS, wrapT;
    float    alphaFactor;      // baseColorFactor.w
    float    alphaCutOff;
    uint32_t pad[2];
    DMatLevel levels[CHORD_MAX_TEX_LEVELS];
};
// extension of a masked triangle's 48-byte record, in the TWO slots behind it.  Everything a row unit of the tile kernel needs to
// sample the triangle's alpha is in here -- the chosen level's first byte and size, the wraps, the material's factor and cut-off --
// so a unit's set-up is one round trip (this record), not three dependent ones (extension -> material -> level offsets).
struct __attribute__((aligned(16))) TriRecMaskExt {         // 96 B, 76 used
    float    uw[3], vw[3], iw[3];   // u / w, v / w, 1 / w per vertex
    uint32_t levelFilter;      // level | linear << 8
    uint32_t levelBase;        // first alpha byte of that level in dTexAlpha; 0xFFFFFFFF: white fallback (alpha 1)
    uint32_t dims;             // (width - 1) | (height - 1) << 16 of that level
    uint32_t wraps;            // wrapS | wrapT << 16 (CHORD_WRAP_*: 16-bit values)
    float    alphaFactor, alphaCutOff;
    uint32_t magicS, biasS, magicT, biasT;   // DMatLevel's, for the remainders of non-power-of-two sizes
    uint32_t pad[5];
};
#define CHORD_MASK_EXT_SLOTS 2u    // record slots an extension takes (a masked triangle: 1 + CHORD_MASK_EXT_SLOTS)

// One flattened (object, group) instance with everything the group cull needs to START at the records it tests: the reference's
// thread walks object -> primitive -> group -> group indices -> meshlets (instance_culling.hlsl:133-208: five dependent fetches);
// the walk is the same for every frame, so it is done once at upload.  24 B.
struct DGroupRef {
    uint32_t object;               // owner
    uint32_t group;                // index into dGroups | meshlet count << 28 (<= 4)
    uint32_t meshlet[4];           // global meshlet ids of the group's (up to) four meshlets; unused slots repeat the first
};

struct DObjFrame {             // per object, per frame (written by the object-cull kernel), 208 B
    float    mvp[16];          // row-major VP * M            (mesh_raster.hlsl:90-91)
    float    mvpLast[16];      // row-major VP_last * M_last  (hzb_mainview_culling.hlsl:77-83)
    float    localToView[12];  // rows 0..2 of V * M          (instance_culling.hlsl:170)
    float    camLS[3];         // mul(translatedWorldToLocal, (0,0,0,1)).xyz (nanite_shared.hlsli:65)
    float    maxScale;         // scaleExtractFromMatrix.w
    uint32_t visible;
    uint32_t isOrtho;          // mul(VP, M)[3][3] == 1 (base.hlsli:243-246)
    uint32_t pad[2];
};

struct DView {                 // constants of one frame
    ChordCameraView          view;
    ChordInstanceCullingView iv;
    uint32_t                 flags;
    uint32_t                 width, height;
    uint32_t                 pad;
};

// Multi-GPU screen ownership ("screen tiles (e.g. 64x64) ... visbuffer stored rank-major"): the unit is the rasterizer's
// 64x64-pixel tile, every tile has one owner, the map is a table (any assignment: compact regions, load-balanced, interleaved).
// A rank's tiles are stored contiguously, tile-linear (64 rows of 64 words each), in the rank's chunk of the visibility buffer:
// tile t lives in slot tileSlot[t] = owner * slotsPerRank + (index among the owner's tiles), so ONE in-place all-gather of
// slotsPerRank * 4096 words per rank reassembles the frame and a de-tile kernel restores row-major.  The HZB texels a tile
// owns (mips 0..5: one 64x64 tile is exactly one mip-5 texel) travel in slots of the same numbering (below).
struct ShardInfo {
    uint32_t ranks, rank, slotsPerRank, tilesX;        // slotsPerRank: slots of one rank's chunk = the largest tile count of the current map
    const unsigned long long* ownedRows;   // [tilesY] bit tx set <=> this rank owns tile (tx, ty)   (tilesX <= 64: renderer.h:52-53 caps the render size at 4096)
    const uint32_t* tileSlot;              // [tilesX * tilesY] rank-major slot of every tile
};
// Tile slots a context allocates per rank beyond ceil(tiles / ranks), in thousandths: room for a load-balanced map to give a rank
// of light tiles more of them (tile_layout.cpp).  What an all-gather moves is ranks x slotsPerRank slots, slotsPerRank = the
// largest tile count of the CURRENT map -- the default map uses none of the slack.
#define CHORD_TILE_SLACK_PERMILLE 250u
// HZB texels of one tile, level by level: level l is (32 >> l)^2 texels at offset 1365 - (1365 >> 2l)  (0, 1024, 1280, 1344, 1360, 1364)
#define CHORD_HZB_TILE_TEXELS 1365u
#define CHORD_HZB_SLOT_HALVES 1408u          // a tile's slot in the mid-frame exchange buffer: the min chain's texels, padded
// a tile's slot in the end-of-frame exchange buffer: min texels | max texels | {valid-range min, max, bin entries of the frame, 0} (uint32) | padding
#define CHORD_HZB_FINAL_SLOT_HALVES 2832u
#define CHORD_HZB_FINAL_MAX_OFFSET 1408u
#define CHORD_HZB_FINAL_RANGE_OFFSET 2816u   // (halves; 4-byte aligned)
#ifdef __HIPCC__
__device__ __forceinline__ uint32_t hzb_slot_level_offset(uint32_t l) { return CHORD_HZB_TILE_TEXELS - (CHORD_HZB_TILE_TEXELS >> (2u * l)); }
__device__ __forceinline__ bool shard_owns_tile(const ShardInfo& s, uint32_t tx, uint32_t ty) { return (s.ownedRows[ty] >> tx) & 1ull; }
// does the rank own a tile of the rectangle [tx0, tx1] x [ty0, ty1]?  (small triangles and clusters: one or two rows)
__device__ __forceinline__ bool shard_owns_any_tile(const ShardInfo& s, uint32_t tx0, uint32_t ty0, uint32_t tx1, uint32_t ty1)
{
    const unsigned long long cols = ((2ull << (tx1 - tx0)) - 1ull) << tx0;
    unsigned long long any = 0ull;
    for (uint32_t ty = ty0; ty <= ty1; ty++) any |= s.ownedRows[ty] & cols;
    return any != 0ull;
}
#endif

// Work lists between the raster kernels (device memory, counts in DeviceCounters)
struct TriRec {                // 48 B: one set-up triangle (snapped 24.8 vertices, vertex depths, id)
    int32_t  X[3]; int32_t Y[3];
    float    d[3];
    uint32_t payload;
    uint32_t twoSided;         // bit 0 two-sided, bit 1 orientation sign of the snapped triangle, bit 2 masked: a TriRecMaskExt follows
    uint32_t pad;              // 1 / float(2A)
};
// 32 B: the same for a triangle whose vertices are at most 64 px apart (deltas fit 16 bits) -- nearly all of them.
// Orientation sign and 1/2A are recomputed by the consumer from the deltas (the same expressions, exact integers).
// A bin entry names a record by index; bit 31 set = the 48-byte list, clear = this one.
struct TriRecC {
    int32_t  X0, Y0;
    int16_t  dX1, dY1, dX2, dY2;
    float    d[3];
    uint32_t payload;
};
#define CHORD_REC_WIDE 0x80000000u
// Pixel blocks (kernels_raster.hip "small clusters"): a cluster whose emitted triangles all fall into a window of
// CHORD_BLOCK_WIN x CHORD_BLOCK_WIN pixels is resolved by its setup wave in LDS and leaves the kernel as one dense block of
// packed visibility words per tile it touches instead of one record + bin entry per triangle.  A block lives in the
// block pool at a 16-byte granule offset: word 0 = header (x0 | y0 << 6 | (w-1) << 12 | (h-1) << 16, tile-local, and
// ceil(65536 / w) in the high half), then w x h words, row-major.  Its bin entry is CHORD_REC_BLOCK | granule offset.
#define CHORD_REC_BLOCK 0xC0000000u
#define CHORD_REC_INDEX_MASK 0x3FFFFFFFu
// A bin entry that names the 48-byte record of an alpha-tested (masked) triangle: CHORD_REC_WIDE with bit 29 set.  Such entries are
// scan-converted by a pass of their own (raster_masked_tile_kernel) and skipped by the tile kernel, which therefore needs to look at
// nothing but the bin word.  Indices of the 48-byte list stay below 2^29 (chordvis_set_limits caps the list at 0x7FFFFFC0 / 4 records).
#define CHORD_REC_MASKED 0xA0000000u
#define CHORD_REC_WIDE_INDEX 0x1FFFFFFFu
#define CHORD_BLOCK_WIN 16
struct ClipTri { uint32_t objectId, meshletId, slot, tri; };    // needs the homogeneous clipper (the draw command rides along: the setup kernels read different lists)

// The record list is cut into LIST_SHARDS independent sub-lists (own counter, own region) so that
// list allocation is not serialised on one memory-side atomic (one word sustains only ~88 returning
// atomics/us on MI355X); a wave picks its shard from its global wave id.
#define CHORD_LIST_SHARDS 64u
#define CHORD_SHARD_STRIDE 16u
struct DeviceCounters {
    // shard counters sit one per 64-byte line (CHORD_SHARD_STRIDE words apart): atomics on different words of one
    // line still serialise at the L2
    uint32_t triCount[CHORD_LIST_SHARDS * CHORD_SHARD_STRIDE];   // 48-byte records appended this frame (both raster passes)
    uint32_t triCountC[CHORD_LIST_SHARDS * CHORD_SHARD_STRIDE];  // 32-byte records
    uint32_t clipTriCount[2];                   // per raster pass
    uint32_t largeCount[2][CHORD_LIST_SHARDS * CHORD_SHARD_STRIDE];  // per raster pass and list shard: records touching more than 2x2 tiles
    uint32_t overflow;                          // bit0 record list / tile bin / large list, bit1 clip list, bit2 bin chunk wait timed out
    uint32_t binPoolCount[2];                   // per raster pass: overflow chunks handed out
    uint32_t pad;
    uint32_t blockGranules[CHORD_LIST_SHARDS * CHORD_SHARD_STRIDE];   // 16-byte granules of the block pool handed out this frame, per shard
    uint32_t pad3[10];                          // (the four totals below + FrameState::listCounts are ONE 64-byte line: frame_cull_fused_kernel's last workgroup writes it whole)
    // triangles (meshlet triangle counts) of the commands each list producer emitted this frame
    unsigned long long trisInstanceCulled, trisHzbVisible0, trisHzbVisible1, pad2;
};

// Everything a frame zeroes lives in ONE allocation so the frame starts with one memset:
// counters, the four command-list counts, and the per-pass tile bin counts (2 x tiles follow).
#define CHORD_BIN_CHUNK_SHIFT 10
#define CHORD_BIN_CHUNK (1u << CHORD_BIN_CHUNK_SHIFT)   // entries per overflow chunk
#define CHORD_BIN_MAX_CHUNKS_DEFAULT 240u            // overflow chunks per tile: bins hold up to binCap + 240 Ki entries (chordvis_set_limits)
#define CHORD_BIN_MAX_CHUNKS_LIMIT 3072u
#define CHORD_TILE_MAX_SLICES 64u                    // a long bin is cut into at most this many slices
#define CHORD_BIN_CHUNK_INVALID 0xFFFFFFFFu
#define CHORD_TILE_SLICE_SHIFT 11                    // long bins are scan-converted in slices of 2048 entries
#define CHORD_HOT_TILES 32u                          // hot tiles (bins beyond 65 536 entries) the tile schedule remembers per pass for the next frame's block kernel
#ifndef CHORD_TILE_SHIFT
#define CHORD_TILE_SHIFT 6                       // log2 of the raster tile side in pixels (5 or 6)
#endif
#define CHORD_TILE (1 << CHORD_TILE_SHIFT)
#define CHORD_MAX_TILES ((4096u >> CHORD_TILE_SHIFT) * (4096u >> CHORD_TILE_SHIFT))   // renderer.h:52-53 caps the render size at 4096^2
#if CHORD_TILE_SHIFT == 6
#define CHORD_TILECOUNT_STRIDE 16u               // one bin counter per 64-byte line
#ifndef CHORD_BIN_CAP
#define CHORD_BIN_CAP 16384u
#endif
#else
#define CHORD_TILECOUNT_STRIDE 4u                // four bin counters per 64-byte line
#define CHORD_BIN_CAP 8192u
#endif
static_assert(CHORD_BIN_CAP <= (1u << 20), "tile x bin capacity is formed with a 24-bit multiply in 32 bits (bin_put)");
struct FrameState {
    DeviceCounters counters;
    uint32_t listCounts[8];        // [0..3] command lists of the frame, [4] this rank's share of list 0 (sharded: written by the group cull), [5] this rank's clusters of a foreign list (stripe filter), [6 + pass] clusters a dense launch's block kernel left over
    uint32_t tileCount[2 * CHORD_MAX_TILES * CHORD_TILECOUNT_STRIDE];   // pass p starts at p * tiles * stride
};

static_assert(offsetof(FrameState, counters.trisInstanceCulled) % 64 == 0 && offsetof(FrameState, listCounts) == offsetof(FrameState, counters.trisInstanceCulled) + 32 &&
              offsetof(FrameState, tileCount) == offsetof(FrameState, listCounts) + 32, "triangle totals + list counts: one 64-byte line (frame_cull_fused_kernel)");

struct CmdList {
    uint32_t*     count = nullptr;
    ChordDrawCmd* cmds = nullptr;
    uint32_t      capacity = 0;
    ChordCountAndCmd handle() const { return ChordCountAndCmd{count, cmds, capacity}; }
};

struct HzbBuffers {
    ChordHZBDesc desc{};
    uint16_t* minTexels = nullptr;
    uint16_t* maxTexels = nullptr;
    uint32_t* validRange = nullptr;
    bool valid = false;
    ChordHZB handle() const { return ChordHZB{desc, minTexels, maxTexels, validRange}; }
};

// GPU timestamp tags: a stamp closes the segment that started at the previous stamp.
enum StampTag { S_FRAME_BEGIN = 0, S_CLEAR, S_CULL, S_HZBCULL, S_R_CLUSTER, S_R_CLIP, S_R_CHUNK, S_STAGE0_END,
                S_HZB0, S_STAGE1_END, S_HZBF, S_OTHER,
                S_EXCH_HZB, S_EXCH_VIS,     // sharded frames: the segment is the mid-frame HZB exchange / the visibility all-gather
                S_EXCH_CULL,                // ... the all-gather of the group cull's rank masks (sharded cull)
                S_EXCH_FINAL };             // ... the small end-of-frame exchange (library-run frames stamp it apart from the image gather)

} // namespace chord

struct ChordCtx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool ownStream = false;
    std::string lastError;
    int numCUs = 256;

    // scene
    bool sceneLoaded = false;
    uint32_t objectCount = 0, primCount = 0, materialCount = 0;
    uint32_t meshletCount = 0, groupCount = 0;
    uint32_t groupInstances = 0;      // sum over objects of their primitive's group count
    uint32_t cmdCapacity = 0;         // sum over objects of their primitive's meshlet count (all LODs)
    uint64_t instTriangles = 0;       // triangles of every meshlet instance (sizes the record lists)
    chord::DPrim* dPrims = nullptr;
    chord::DGroup* dGroups = nullptr;
    chord::DMeshlet* dMeshlets = nullptr;
    uint32_t* dGroupIndices = nullptr;
    uint32_t* dMeshletData = nullptr;
    float* dPositions = nullptr;
    chord::DObjStatic* dObjStatic = nullptr;
    chord::DMaterial* dMaterials = nullptr;   // per material: what the masked buckets sample
    uint8_t* dTexAlpha = nullptr;             // alpha channel of every level of every texture, back to back
    float* dTexcoords = nullptr;              // float2 per vertex (textureCoord0Buffer), or null
    bool anyMasked = false;
    chord::DGroupRef* dGroupRefs = nullptr;   // per flattened (object, group) instance (static: the object -> primitive binding is the scene's)
    chord::DBVHNode* dBvhNodes = nullptr;   // every primitive's tree (or null: the scene came without)
    bool bvhComplete = false;         // every primitive has a validated tree
    int cullMode = 0;                 // 0 flat (the reference's dispatch), 1 hierarchical (chordvis_set_cull_mode)
    // depth-only views (shadow cascades): a child context of the cascade size that shares this context's scene buffers
    ChordCtx* depthCtx = nullptr;
    bool sharedScene = false;         // this IS such a child: the scene buffers belong to the parent
    std::vector<float*> dDepthImages; // one D32 image per view (cascadeDim^2 floats)
    uint32_t depthDim = 0;
    bool shadowHistoryValid = false;  // CascadeShadowHistory (mesh_raster.cpp:548-563): config + light direction of the cached cascades
    ChordCascadeConfig shadowHistoryConfig{};
    float shadowHistoryDir[3] = {0, 0, 0};
    int depthViewCurrent = -1;        // (child) the view whose object matrices dObjFrame holds
    float* depthOutTarget = nullptr;  // (child) set around a depth pass: the fused tile-out writes this D32 image directly
    int fusedDepthView = -1;          // (parent) the view whose HZB mips 0..5 the child's chain 0 holds, straight out of its depth pass
    std::vector<ChordInstanceCullingView> instanceViews;   // chordvis_set_instance_views (cascadeViewInfos)
    ChordObject* dObjectsOwned = nullptr;
    const ChordObject* dObjects = nullptr;
    std::vector<chord::DPrim> hPrims;
    std::vector<chord::DObjStatic> hObjStatic;

    // frame
    chord::DView hView{};
    chord::DView* dView = nullptr;
    bool viewSet = false;
    bool viewDirty = false;            // hView not yet on the device (object_cull publishes it from its kernel argument)
    bool zeroFrameStateInCull = false; // object_cull zeroes the FrameState block instead of a memset
    size_t frameStateZeroBytes = 0;
    chord::DObjFrame* dObjFrame = nullptr;
    uint8_t* dGroupMask = nullptr;
    uint32_t* dBlockCounts = nullptr;
    uint32_t cullBlocks = 0;

    // command lists: 0 = post instanceCulling, 1 = hzb visible, 2 = hzb rejected
    chord::CmdList lists[3];
    ChordDrawCmd* dRankCmds = nullptr; // sharded frames: the commands of a raster pass whose clusters touch this rank's rows
    ChordDrawCmd* dMineCmds = nullptr; // sharded frames: the post-instanceCulling commands of THIS rank's clusters, written by the group cull itself (count: listCounts[4])
    bool mineValid = false;            // ... produced by the last chordvis_instance_culling
    bool lazyFullList = false;         // (set around the cull of a sharded frame inside the library) the group cull writes only the rank's list
    bool fullListStale = false;        // ... lists[0] has not been written for the last cull: launch_full_list makes it from the cull's masks
    bool listMine[3] = {false, false, false};   // lists[k] currently holds only this rank's clusters (it was culled from the rank's list)
    // sharded group cull (chordvis_frame_phase_cull): a rank tests only its range of count blocks and writes, per group instance, one
    // word = for each of the group's <= 4 meshlets the 8-bit set of ranks whose tiles the cluster touches (0: culled); the words of all
    // ranks are all-gathered, and prefix + scatter run locally from them
    uint32_t* dCullExchange = nullptr; // [ranks][cullChunkBlocks * 256 words | cullChunkBlocks triangle sums]
    uint32_t cullChunkBlocks = 0;      // count blocks per rank = ceil(cullBlocks / ranks)
    uint32_t cullExchangeRanks = 0;    // the rank count the buffer was made for
    uint8_t* dTileOwner = nullptr;     // sharded: owner of every tile (what the rank masks are made from)
    bool cullPhaseDone = false;        // this frame's chordvis_frame_phase_cull ran: phase a unpacks the exchanged words instead of culling
    bool exchangeSlotsFresh = false;   // this frame's fused tile kernel wrote the rank's slots of the end-of-frame exchange buffer (phase c needs them)
    ChordDrawCmd* dLeftCmds = nullptr; // dense launches: the clusters the block kernel left to the record kernel (kernels_raster.hip)
    uint32_t* dCounts = nullptr;      // 4 x u32 backing the list counts

    // gbuffer
    uint32_t width = 0, height = 0;
    chord::ShardInfo shard{1, 0, 0, 0, nullptr, nullptr};
    uint32_t slotCapacity = 0;            // sharded: tile slots allocated per rank (chordvis_tile_slot_capacity), >= shard.slotsPerRank
    std::vector<uint8_t> tileOwners;      // sharded: owner of every tile (the same table on every rank); empty = the default layout of configure_targets
    bool tileOwnersExplicit = false;      // ... set by chordvis_set_tile_owners / chordvis_rebalance (kept across allocate_gbuffer of the same size)
    unsigned long long* dShardTables = nullptr;   // [64] ownedRows, then [tiles] tileSlot (uint32)
    uint32_t* dTileLoads = nullptr;       // [tiles] bin entries per tile of the last frame, every rank's tiles (written when the end-of-frame exchange is unpacked)
    void* comm = nullptr;             // ncclComm_t of a one-process-per-GPU host (chordvis_comm_init_rank), or null
    // pipelined frames over RCCL (chordvis_comm_set_pipelined): a second communicator carries the image of frame i beside frame i + 1
    void* commBulk = nullptr;
    hipStream_t commResolveStream = nullptr;
    hipEvent_t commPhaseB = nullptr, commVisReady[2] = {nullptr, nullptr};
    bool commPipelined = false;
    uint64_t commFrameSerial = 0;
    uint64_t* dVis = nullptr;         // in use (owned or caller's)
    uint64_t* dVisOwned = nullptr;
    uint64_t* dVisResolved = nullptr; // row-major copy when ranks > 1
    uint64_t visWords = 0;
    bool visExternal = false;

    // HZB: slot 0 temp, 1/2 history ping-pong
    chord::HzbBuffers hzb[3];
    int historySlot = 0;              // 0 = none, else 1 or 2
    int pendingTailSlot = 0;          // history chain whose mips 6.. + valid range are still to be reduced (carried by the next frame's first kernel)
    uint32_t* dRangePartials = nullptr;   // per mip-0 block {min, max} of valid depth
    uint32_t* dTileRange = nullptr;       // per 64x64 tile {min, max} of valid depth (fused HZB)
    volatile uint32_t* hBinHint = nullptr;    // pinned, device-visible: per raster pass the longest bin of the last frame the GPU finished (hot tiles: launch_raster)
    uint32_t* dBinHint = nullptr;             // ... its device address
    uint32_t* dHotTiles = nullptr;            // [2 passes][1 + CHORD_HOT_TILES]: the hot tiles of the last frame (count, tile | very hot << 31), device memory that no frame zeroes
    bool fuseHzb = false;                 // inside render_frame: the tile kernel emits HZB mips 0..5
    bool fuseHzbTemp = false;             // ... also into the temporary chain (slot 0) for stage 1
    int fuseHzbSlot = 1;                  // history slot being produced this frame
    uint16_t* dHzbExchange = nullptr;      // sharded frames, mid-frame exchange: per tile slot the min chain's mips 0..5 after the first raster pass (CHORD_HZB_SLOT_HALVES)
    uint16_t* dHzbFinalExchange = nullptr; // ... end-of-frame exchange: per tile slot min | max | valid range | bin entries (CHORD_HZB_FINAL_SLOT_HALVES)
    uint64_t hzbFinalExchangeChunkBytes = 0;
    uint64_t* dVisAlt = nullptr;           // ... the visibility words (rank-major) and their row-major copy of the OTHER frame in flight
    uint64_t* dVisResolvedAlt = nullptr;
    hipEvent_t visReadyEvent[2] = {nullptr, nullptr};   // [0]: the resolved image of the last submitted frame is complete, [1]: of the frame before
    uint64_t hzbExchangeHalves = 0, hzbExchangeChunkHalves = 0;

    // raster work lists: triangle records, per-tile bins, clip list
    chord::TriRec* dTris = nullptr;
    uint32_t triCap = 0;               // 48-byte records, all shards together
    chord::TriRecC* dTrisC = nullptr;
    unsigned long long* dBlockPool = nullptr;   // pixel blocks of small clusters (CHORD_REC_BLOCK), [CHORD_LIST_SHARDS][blockCap] granules of 16 bytes
    uint32_t blockCap = 0;             // granules per shard
    uint32_t triCapC = 0;              // 32-byte records, all shards together
    chord::FrameState* dFrameState = nullptr;
    uint32_t* dTileBins = nullptr;     // [2 passes][tiles][binCap]: the first binCap entries of every tile's bin
    // entries beyond binCap live in CHORD_BIN_CHUNK-entry chunks handed out from a pool; the j-th overflow chunk of
    // a tile is named by dBinChunkTab[pass][tile][j] = raster serial << 32 | chunk id (never zeroed: stale serials
    // do not match)
    uint32_t* dBinPool = nullptr;      // [2 passes][binPoolChunks][CHORD_BIN_CHUNK]
    unsigned long long* dBinChunkTab = nullptr;
    uint32_t binPoolChunks = 0;        // per pass
    uint32_t binMaxChunks = CHORD_BIN_MAX_CHUNKS_DEFAULT;   // per tile
    // capacities a host may raise before upload_scene / allocate_gbuffer (chordvis_set_limits)
    uint64_t limitRecords = 64ull << 20;
    uint32_t limitPoolChunks = 32768;
    uint32_t rasterSerial = 0;
    bool inFrame = false;              // inside render_frame / frame_phase_*: per-pass counts were zeroed at frame begin
    uint32_t binCap = 0, tilesX = 0, tilesY = 0;
    chord::ClipTri* dClipTris = nullptr;
    uint32_t clipTriCap = 0;
    uint32_t* dTileMarker = nullptr;   // [markerDim.y][markerDim.x] uint4: shading types present per 8x8 pixels
    uint32_t* dShadingTiles = nullptr; // [markerDim.x * markerDim.y] uint2 + {count, pad, uint4 dispatch args} behind them
    uint32_t* dTileOrder = nullptr;    // [1 + tileItemCap]: item count, then work items heaviest first
    uint32_t* dTileOrderKeep = nullptr; // the same for the first pass of main-view frames on one GPU, kept across frames (launch_raster)
    uint32_t orderAge = 0xFFFFFFFFu;   // frames since dTileOrderKeep was made; 0xFFFFFFFF: not valid (new target, scene, map, switches)
    uint32_t* dTileOrderKeep1 = nullptr; // ... and of a frame's HEAVY second pass (every tile listed, touched or not: launch_raster)
    uint32_t orderAge1 = 0xFFFFFFFFu;
    uint32_t orderFlip[2] = {0u, 0u};  // which half of dTileOrderKeep / dTileOrderKeep1 this frame reads (the other one is being made for the next)
    uint32_t orderKeepFrames = 1u;     // chordvis_set_tile_schedule_keep: != 0 -- a pass runs under the schedule the frame before made for it (launch_raster: tileOrderNext); 0: a schedule kernel in every pass.  (CHORDVIS_TILE_NEXT=0, A/B runs: frames a schedule-kernel schedule is reused for)
    unsigned long long* dTileSlabs = nullptr;   // [tiles][TILE*TILE]: where the slices of a split tile meet (all zero between uses)
    uint32_t tileItemCap = 0;
    uint32_t* dLargeList = nullptr;    // [2 passes][CHORD_LIST_SHARDS][largeCap / 2 / CHORD_LIST_SHARDS] record indices
    uint32_t largeCap = 0;
    chord::DeviceCounters* dCounters = nullptr;
    bool pendingClear = false;         // the next raster pass starts every tile from zero (fused clear)
    // depth-only pass state (renderMeshDepth, mesh_raster.cpp:159-206); set by chordvis_render_mesh_depth around its raster
    bool depthOnly = false, depthClamp = false;
    float depthBiasConst = 0.0f, depthBiasSlope = 0.0f;

    uint64_t launchCount = 0;          // kernel launches since the context was made (CHORD_LAUNCH)
    uint64_t frameLaunchBase = 0;      // ... at the start of the current frame
    uint32_t lastFrameLaunches = 0;    // launches of the last finished frame
    // timers: mode 0 off, 1 = last frame only, 2 = accumulate until chordvis_stats
    int timers = 0;
    std::vector<hipEvent_t> evPool;
    std::vector<int> stampTags;        // tags of evPool[0 .. stampTags.size())
    uint32_t framesStamped = 0;
    uint32_t timerPeriod = 1;          // stamp every timerPeriod-th frame (an event record costs ~5 us of stream idle time)
    uint32_t frameIndex = 0;
    bool stampThisFrame = false;
    uint32_t rasterCalls = 0;          // renderMesh calls since the last clear
    bool shouldStage1 = false;
    chord::CmdList lastRejected;
    bool hzbTailInCull = false;        // render_frame: the phase-1 HZB cull reduces levels 6.. of the chain itself (no hzb_tail_kernel before it)
    // short scenes on one GPU: instanceCulling and the phase-0 occlusion cull in one kernel (kernels_cull.hip frame_cull_fused_kernel)
    bool fuseCullFrame = false;        // render_frame -> launch_group_cull: this instanceCulling opens a frame of chordvis_render_frame (the fused kernel may run)
    const chord::HzbBuffers* fuseCullHzb = nullptr;   // ... and the history chain the frame's phase-0 cull will test against (NULL: no history / occlusion culling off)
    bool fusedCullDone = false;        // launch_group_cull took the fused path WITH the phase-0 cull: chordvis_hzb_culling(first stage) of this frame launches nothing
    uint32_t cullSerial = 0;           // launch serial of the fused kernel's look-back words
    unsigned long long* dCullLookback = nullptr;   // [numCUs][2]
    uint32_t debugFlags = 0;           // ablation switches for measurements (chordvis_set_debug)
    unsigned long long* dTileClocks = nullptr;   // [2][CHORD_MAX_TILES] per-tile ticks when debug bit 4 is set
};

// every kernel launch of the library goes through here: the context counts them (ChordStats::kernelLaunches -- what a sub-millisecond
// frame is bounded by is its launch count x the launch floor)
#define CHORD_LAUNCH(ctx, ...) do { (ctx)->launchCount++; hipLaunchKernelGGL(__VA_ARGS__); } while (0)

namespace chord {

int fail(ChordCtx* ctx, int code, const char* what, hipError_t e = hipSuccess);
int alloc_scene_work_buffers(ChordCtx* c);

#define CHORD_HIP(ctx, call)                                                         \
    do {                                                                             \
        hipError_t e_ = (call);                                                      \
        if (e_ != hipSuccess) return ::chord::fail((ctx), CHORDVIS_E_HIP, #call, e_); \
    } while (0)

// kernel launchers (implemented in the .hip translation units) ---------------------------------
void launch_group_cull(ChordCtx* c, const CmdList& out);
void launch_full_list(ChordCtx* c);             // the full post-cull list of a sharded frame, when a consumer asks for it
void launch_cull_masks(ChordCtx* c, bool wholeRange = false);   // sharded cull: objects + this rank's range of group instances -> rank-mask words (its chunk of dCullExchange)
bool cull_shardable_config(const ChordCtx* c);  // the sharded cull applies to this configuration: 2..8 ranks, flat cull mode, a scene and a sharded G-buffer (debug bit 524288 -- set alike on every rank -- turns it off)
bool cull_shardable(const ChordCtx* c);         // ... and its exchange buffer stands (made at set-up): what a frame decides its first collective by -- state every rank shares
int ensure_cull_exchange(ChordCtx* c);          // chordvis_abi.cpp: (re)allocates dCullExchange for the current scene / rank count
int prepare_cull_exchange(ChordCtx* c);         // ... at set-up time (upload_scene, set_shard / allocate_gbuffer), wherever the sharded cull can apply
void launch_hzb_cull(ChordCtx* c, const HzbBuffers& hzb, int phase, const CmdList& in, const CmdList& outVisible,
                     const CmdList* outRejected);
hipError_t launch_raster(ChordCtx* c, const CmdList& in, bool clearTiles);   // first failing HIP call, or hipSuccess
void launch_hzb_cull_generic(ChordCtx* c, const HzbBuffers& hzb, const ChordInstanceCullingView& iv, const float rel[3], float extentScale,
                             bool useLastFrame, const CmdList& in, const CmdList& out);
void launch_depth_extract(ChordCtx* c, const unsigned long long* vis, float* depth, size_t words);              // high word of every visibility word
void launch_depth_expand(ChordCtx* c, const float* depth, unsigned long long* vis, size_t words);               // and back: depth << 32
void launch_hzb_build(ChordCtx* c, HzbBuffers& out, bool bMin, bool bMax, bool bValidRange);
void launch_hzb_tail(ChordCtx* c, HzbBuffers& out, bool bMax, bool bValidRange);   // mips 6.. + range from per-tile partials
void launch_detile(ChordCtx* c, hipStream_t stream = nullptr);
void launch_hzb_untile(ChordCtx* c, HzbBuffers& out, bool finalChain);   // exchanged tile slots -> mips 0..5 of the chain (final: min + max + per-tile ranges and loads)
void launch_rank_filter(ChordCtx* c, const CmdList& in, const CmdList& out);
int tile_layout(uint32_t tilesX, uint32_t tilesY, uint32_t ranks, const uint32_t* loads, uint32_t cap, uint8_t* owners);   // tile_layout.cpp
int install_tile_owners(ChordCtx* c);                                     // chordvis_abi.cpp: c->tileOwners -> device tables
void launch_visibility_mark(ChordCtx* c, const unsigned long long* vis, const ChordDrawCmd* cmds, const uint32_t* cmdCount, uint32_t* marker);
void launch_shading_tiles(ChordCtx* c, const uint32_t* marker, uint32_t shadingType, uint32_t* tiles, uint32_t* count, uint32_t* args);
void stamp(ChordCtx* c, int tag);               // no-op when timers are off
int comm_render_frame(ChordCtx* c);             // multi_gpu.cpp: phase a -> ncclAllGather -> phase b -> ncclAllGather -> phase c

} // namespace chord
