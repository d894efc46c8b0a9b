/*
 * chordvis.h — C ABI of the MI355X-native visibility hot path (libchordvis.so).
 *
 * The reference has no plugin/FFI boundary for this path: its boundary is the
 * free-function pass surface of source/renderer (gltf_rendering.h:37-108,
 * postprocessing.h:41-54) called from DeferredRenderer::render
 * (renderer.cpp:319-345), sitting on the Vulkan device layer of
 * source/graphics.  Every entry point below names the reference interface it
 * replaces.  Signatures carry plain pointers and sizes only; device memory is
 * named by raw device pointers (the counterpart of the reference's bindless
 * uint32 ids obtained from asSRV/asUAV, render_helper.h:290-359).
 *
 * Conventions
 *  - every function returns 0 on success, a negative CHORDVIS_E_* code otherwise;
 *    chordvis_last_error(ctx) returns the message (reference: check()/
 *    checkVkResult() assert + log, utils.h:57-72, graphics/common.h:337).
 *  - one host thread per context; all passes are enqueued on the context's HIP
 *    stream in call order and never synchronize the host (reference: one
 *    graphics queue, one vkQueueSubmit per frame, command_list.cpp:107-153).
 *  - handles (ChordCountAndCmd, ChordHZB) point into context-owned device
 *    memory and stay valid until the next chordvis_instance_culling /
 *    chordvis_build_hzb into the same slot (reference: pooled buffers recycled
 *    after N frames, buffer_pool.cpp:71,122).
 */
#ifndef CHORDVIS_H
#define CHORDVIS_H

#include "chordvis_types.h"

#ifdef __cplusplus
extern "C" {
#endif
/* libchordvis.so is built with -fvisibility=hidden: the entry points declared in this header are its ONLY dynamic symbols (the
 * library's C++ internals live in a namespace `chord`, which is also the reference's -- a host must never see them). */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define CHORDVIS_OK              0
#define CHORDVIS_E_INVALID      -1   /* bad argument / call order            */
#define CHORDVIS_E_HIP          -2   /* a HIP runtime call failed            */
#define CHORDVIS_E_NO_DEVICE    -3   /* no usable gfx950 device              */
#define CHORDVIS_E_CAPACITY     -4   /* scene exceeds a documented limit     */
#define CHORDVIS_E_COMM         -5   /* RCCL missing / a collective failed   */

typedef struct ChordCtx ChordCtx;

/* CountAndCmdBuffer — postprocessing.h:48.  Device pointers. */
typedef struct ChordCountAndCmd {
    uint32_t*     count;     /* device: number of valid commands           */
    ChordDrawCmd* cmds;      /* device: uint3 commands                     */
    uint32_t      capacity;  /* commands the buffer can hold               */
} ChordCountAndCmd;

/* HZBContext — render_helper.h:411-451.  Device pointers into one f16 chain per channel. */
typedef struct ChordHZB {
    ChordHZBDesc desc;
    uint16_t*    minTexels;   /* device, desc.totalTexels halves, or NULL    */
    uint16_t*    maxTexels;   /* device or NULL                              */
    uint32_t*    validRange;  /* device uint2 {asuint(min), asuint(max)} or NULL */
} ChordHZB;

/* Work-list capacities (the counterpart of the reference's pool sizes); zero fields keep the default.  Call before
 * chordvis_upload_scene / chordvis_allocate_gbuffer.  The record lists are sized FROM THE SCENE at upload
 * (2 x its instanced triangles + 256 Ki, chordvis_abi.cpp alloc_scene_work_buffers) and capped by these limits; the default
 * caps fit BASELINE configs 1-4 and config 5 with pixel blocks; config 5 in the record form (1 G sub-pixel triangles in one
 * pass, debug only) needs ~1.1 G records and ~1.1 M pool chunks per pass. */
typedef struct ChordLimits {
    uint64_t maxTriangleRecords;   /* upper bound of the 32-byte record list (the 48-byte list gets a quarter, at least 1 Mi); default cap 64 Mi */
    uint32_t binPoolChunks;        /* 1024-entry overflow chunks per raster pass; default 32 Ki */
    uint32_t binMaxChunksPerTile;  /* default 240, at most 3072 */
} ChordLimits;

/* VisibilityTileMarkerContext — visibility_tile.h:11-19.  One uint4 (128 shading-type bits) per 8x8 pixels. */
typedef struct ChordTileMarker {
    uint32_t* marker;            /* device, 4 words per texel, markerDim[0] x markerDim[1] texels, row-major */
    uint32_t  visibilityDim[2];
    uint32_t  markerDim[2];      /* ceil(visibilityDim / 8), visibility_tile.cpp:31 */
} ChordTileMarker;

/* VisibilityTileContxt — visibility_tile.h:29-33 */
typedef struct ChordShadingTiles {
    uint32_t* tileCmd;           /* device uint2 per tile: pixel origin of an 8x8 tile; order unspecified (as in the reference) */
    uint32_t* count;             /* device: number of tiles */
    uint32_t* dispatchIndirect;  /* device uint4 {(count + 3) / 4, 1, 1, 1} */
    uint32_t  capacity;          /* markerDim[0] * markerDim[1] */
} ChordShadingTiles;

/* Camera inputs of ICamera (camera.h:22-139) + ViewportCamera::updateMatrixMisc (viewport.cpp:434-445). */
typedef struct ChordCameraDesc {
    double   position[3];
    double   front[3];
    double   worldUp[3];
    float    fovy;          /* radians */
    float    jitter[2];     /* PerframeCameraView::jitterData.xy, pixels in (-0.5, 0.5) */
    double   zNear;
    double   zFar;
    uint32_t width, height;
} ChordCameraDesc;

/* GPU timestamps — the labels DeferredRenderer::render inserts (renderer.cpp:322-344). */
typedef struct ChordStats {
    float    msClear;              /* "Clear GBuffers"                   */
    float    msInstanceCulling;    /* "GLTF Instance Culling"            */
    float    msStage0;             /* "GLTF Visibility Stage0"           */
    float    msHzbStage0;          /* "BuildHZB Post Prepass Stage0"     */
    float    msStage1;             /* "GLTF Visibility Stage1"           */
    float    msHzbFinal;           /* "BuildHZB"                         */
    float    msFrame;              /* clear .. final HZB                 */
    float    msRasterCluster;      /* sum over the frame's raster_setup_kernel launches (per-meshlet setup + binning) */
    float    msRasterClip;         /* ... raster_clip_and_bin_large_kernel + raster_tile_order_kernel                 */
    float    msRasterChunk;        /* ... raster_tile_kernel (per-tile resolve in LDS, tile-out, fused HZB mips 0-5)  */
    uint32_t framesTimed;          /* frames the ms* fields are averaged over               */
    uint32_t rasterLaunches;       /* renderMesh calls this frame (1 or 2)                  */
    uint32_t overflow;             /* non-zero: a deferred raster list overflowed (results invalid) */
    uint32_t countInstanceCulled;  /* commands after instanceCulling     */
    uint32_t countStage0Visible;
    uint32_t countStage0Rejected;
    uint32_t countStage1Visible;
    uint64_t trianglesSubmitted;   /* sum of meshlet triangle counts of rastered commands */
    uint64_t triangleRecords;      /* set-up triangles that survived the per-triangle culls (this frame)  */
    uint64_t binEntries;           /* (triangle, 64x64 tile) pairs binned (this frame, both raster passes) */
    uint32_t tilesTouched[2];      /* 64x64 tiles with at least one bin entry, per raster pass            */
    uint64_t triangleRecordsCompact; /* of triangleRecords, those in the 32-byte form (the rest take 48 bytes) */
    uint64_t pixelBlockBytes;      /* bytes of pixel blocks emitted for small clusters in place of records (this frame); their
                                      triangles are not in triangleRecords, a block counts as one bin entry per tile */
    uint64_t pixelBlocks;          /* number of those blocks (= their bin entries; on frames with hot tiles, plus the bin slots the block
                                      kernel drew ahead and left empty -- a few per wave and hot tile) */
    float    msExchangeHzb;        /* sharded frames: stream time between phase a and phase b = the all-gather of the HZB mip-0 exchange buffer */
    float    msExchangeVis;        /* ... between phase b and phase c = the all-gather of the visibility words (incl. waiting for the slowest rank) */
    float    msExchangeCull;       /* ... between phase cull and phase a = the all-gather of the group cull's rank masks (sharded cull) */
    float    msExchangeFinal;      /* ... library-run frames (chordvis_comm_*, ChordGroup): the small end-of-frame exchange, stamped apart from the image
                                      gather (msExchangeVis is then the image alone); 0 when the host drives the phases */
    uint32_t kernelLaunches;       /* kernel launches of the last finished frame (a sub-millisecond frame is bounded by launches x launch floor) */
    uint32_t largeRecords[2];      /* per raster pass: records touching more than 2 x 2 tiles (binned by the large-record binner, or tested by the tiles themselves) */
    uint32_t clipTriangles[2];     /* per raster pass: triangles that went through the homogeneous clipper */
    float    stampsPerFrame;       /* event records per stamped frame behind the ms* fields (0: no timers).  A record between two kernels keeps the
                                      second from being dispatched under the first: every stamped interval is a few microseconds longer than its
                                      kernels, a stamped frame that many x this longer than an unstamped one (bench.py: roofline.stamp_cost_us) */
} ChordStats;

/* ------------------------------------------------------------------ host-only (no device needed) */

const char* chordvis_version(void);

/* hzb.cpp:49-63 — extent / mip count / offsets of the HZB chain for a render size. */
int chordvis_hzb_desc(uint32_t srcWidth, uint32_t srcHeight, ChordHZBDesc* out);

/* ICamera::fillViewUniformParameter (camera.cpp:17-78), ICamera::computeRelativeWorldFrustum
 * (camera.cpp:80-154), infiniteInvertZPerspectiveRH_ZO (utils.cpp:186-198) and the main-view
 * InstanceCullingViewInfo fill of DeferredRenderer::render (renderer.cpp:175-263).
 * lastFrame may be NULL (first frame: last-frame matrices = current). */
int chordvis_camera_fill_view(const ChordCameraDesc* camera, const ChordCameraView* lastFrame,
                              ChordCameraView* outView, ChordInstanceCullingView* outInstanceView);

/* cascadeComputeCS (cascade_setup.hlsl:79-372, driven by renderShadow mesh_raster.cpp:417-441): the InstanceCullingViewInfo of
 * every shadow cascade -- log / uniform split of the view range (SDSM-tightened for the realtime cascades when the valid
 * depth range of the last buildHZB is given), bounding sphere, light-space lookAt + reverse-Z ortho projection snapped
 * to whole texels, frustum planes.  Host code: the reference runs it on the GPU only to read the depth range without a
 * readback; validDepthMinMax NULL takes the shader's own no-range branch (:118).  views[] is in/out: cascades whose
 * cache is still valid (isCascadeCacheValid, :8-22) are left untouched. */
int chordvis_cascade_setup(const ChordCascadeConfig* config, const ChordCameraView* view, const ChordInstanceCullingView* mainInstanceView,
                           const float lightDir[3], const uint32_t validDepthMinMax[2], uint32_t tickCount, int bCacheValid,
                           ChordInstanceCullingView* views /* [config->cascadeCount] */);

/* SceneNode::getObjectBasicData (scene_node.cpp:42-90): camera-relative f64 -> f32 transforms.
 * Matrices are glm column-major doubles. */
int chordvis_object_basic_data(const double localToWorld[16], const double prevLocalToWorld[16],
                               const double cameraPos[3], const double cameraPosLast[3],
                               ChordObjectBasicData* out);
/* Batched form: fills objects[i].basicData for i < count (ids/pads untouched).
 * prevLocalToWorld / cameraPosLast may be NULL (static object / camera). */
int chordvis_object_basic_data_batch(uint32_t count, const double* localToWorld /*count*16*/,
                                     const double* prevLocalToWorld, const double cameraPos[3],
                                     const double cameraPosLast[3], ChordObject* objects);

/* ------------------------------------------------------------------ producer of the input format (host-only, offline)
 * NaniteBuilder::build (nanite_builder.cpp:882-921): LOD-0 meshlets, up to 11 rounds of group / merge / simplify / split
 * with the simplification error handed up the DAG, cluster groups of at most 4 meshlets, the 8-wide BVH; packed like
 * asset_gltf_helper.cpp:496-548.  The clusterizer, the graph partition and the simplifier are this library's own (the
 * reference calls meshoptimizer and METIS), so the meshlets differ from the reference's for the same mesh; the format and
 * its invariants are the same (nanite_builder.cpp).  Indices: triangle list; texcoord0 may be NULL. */
typedef struct ChordBuiltAsset ChordBuiltAsset;
int chordvis_nanite_build(const float* positions, uint32_t vertexCount, const uint32_t* indices, uint32_t indexCount,
                          const float* texcoord0, ChordBuiltAsset** outAsset);
/* bounds + normal cone of ONE meshlet as the builder computes them (posMin/posMax, coneAxis, coneCutOff, coneApex of `out`;
 * the reference: meshopt_computeMeshletBounds, nanite_builder.cpp:476-486).  positions: the meshlet's own <= 255 vertices;
 * localTriangles: 3 indices into them per triangle (<= 128 triangles). */
int chordvis_meshlet_bounds(const float* positions, uint32_t vertexCount, const uint8_t* localTriangles, uint32_t triangleCount, ChordMeshlet* out);
/* views into the built arrays (valid until chordvis_free_built_asset): one ChordAssetDesc holding one primitive */
int chordvis_built_asset_desc(const ChordBuiltAsset* asset, ChordAssetDesc* outAsset, ChordPrimitive* outPrimitive, uint32_t* outLodCount);
void chordvis_free_built_asset(ChordBuiltAsset* asset);
/* a flat container for a built asset (the reference: cereal + LZ4 archives, serialize.h:217-320) */
int chordvis_save_asset(const ChordBuiltAsset* asset, const char* path);
int chordvis_load_asset(const char* path, ChordBuiltAsset** outAsset);
/* The reference's own container for an asset's geometry: the GLTFBinary archive (asset_gltf.h:260-300) as saveAsset / loadAsset
 * write and read it (serialize.h:217-320: cereal binary archive, LZ4 block compression when lz4 != 0).  Load yields ONE
 * primitive spanning the file (the per-primitive offsets live in the reference's GLTFAsset, another archive); attributes
 * this path does not read (normals, tangents, ...) are skipped on load and written empty. */
int chordvis_save_gltf_binary(const ChordBuiltAsset* asset, const char* path, int lz4);
int chordvis_load_gltf_binary(const char* path, ChordBuiltAsset** outAsset);

/* ------------------------------------------------------------------ context (graphics::Context, graphics.h:88-345) */

/* hipStream: an existing hipStream_t to enqueue on (e.g. the caller's current stream), or NULL for a context-owned
 * NON-BLOCKING stream.  NULL is "no stream given", not "the null stream": work the host enqueues on the legacy default
 * stream is NOT ordered against a context-owned stream.  A host that issues its own collectives or copies between passes
 * hands over the stream it issues them on (PyTorch: a torch.cuda.Stream made current; its default stream has handle 0,
 * which arrives here as NULL), or passes hipStreamLegacy / hipStreamPerThread ((void*)1 / (void*)2), which go through. */
int chordvis_create(int deviceOrdinal, void* hipStream, ChordCtx** outCtx);
int chordvis_destroy(ChordCtx* ctx);
const char* chordvis_last_error(ChordCtx* ctx);
int chordvis_sync(ChordCtx* ctx);

/* GPUScene / asset upload (gpu_scene.h:20-165, asset_gltf.h:278): copies and flattens.  Limits of a scene (CHORDVIS_E_INVALID
 * beyond them): 0x55555555 vertices over all assets (vertex index x 3 is formed in 32 bits; the reference's ByteAddressBuffer
 * offsets are 32-bit BYTE offsets, a third of that), meshlets of at most 255 vertices / 128 triangles, groups of at most four
 * meshlets, alpha-tested textures of at most 16384 x 16384 texels and 15 levels. */
int chordvis_upload_scene(ChordCtx* ctx, const ChordSceneDesc* scene);
/* uploadBufferToGPU("GLTFObjectInfo", ...) renderer.cpp:229 — per-frame object records
 * (count must equal the uploaded scene's objectCount; primitive/material ids must not change). */
int chordvis_update_objects(ChordCtx* ctx, const ChordObject* hostObjects, uint32_t count);
/* Same, for a caller-owned DEVICE array (no copy; must stay valid while bound). */
int chordvis_bind_objects(ChordCtx* ctx, const ChordObject* deviceObjects, uint32_t count);

/* uploadBufferToGPU("PerViewCamera") + ("MainViewInstanceCullingInfo") renderer.cpp:246,262
 * and the r.instanceculling.* cvars -> switchFlags (instance_culling.cpp:22-68). */
int chordvis_set_view(ChordCtx* ctx, const ChordCameraView* view, const ChordInstanceCullingView* instanceView,
                      uint32_t switchFlags);

/* How instanceCulling walks a primitive's cluster groups.  0 (default): flat, one thread per (object, group) like the
 * reference's dispatch (instance_culling.cpp:144-157; it uploads the BVH and never reads it, instance_culling.hlsl:96-99).
 * 1: hierarchical -- resident waves walk every visible object's GPUBVHNode tree (ChordAssetDesc::bvhNodes) and drop a
 * subtree when its sphere, which bounds the parent-error spheres beneath it, already projects below the LOD
 * threshold; only the groups of surviving nodes are tested.  The command list is the same array either way. */
int chordvis_set_cull_mode(ChordCtx* ctx, int hierarchical);

/* Tile schedules of a frame's raster passes (no counterpart: the reference's hardware rasterizer schedules its own tiles).  The first
 * pass writes every tile of the target, so its work items never change; only their order (heaviest bin first) and the cut of long bins
 * depend on the frame.  With `frames` != 0 (default 1) a pass runs under the schedule the SAME pass of the frame before left behind:
 * one workgroup of that frame's tile kernel orders its bin counts for the next frame while the others raster, so no frame launches a
 * schedule kernel and every schedule is exactly one frame old (frames inside chordvis_render_frame / the frame phases; a heavy second
 * pass the same way -- its schedule lists every tile, touched or not).  The first frame after a new target, scene, tile map or switch
 * makes its schedules with the schedule kernel.  0: a fresh schedule from the schedule kernel in every pass.  Order and cut are choices
 * of speed -- the image is the same with any; a host may ignore camera cuts: a stale order costs balance for one frame, never a pixel.
 * (Measured: along a moving camera path a one-frame-old schedule is as good as a fresh one; one reused for 3 / 7 frames -- what a
 * value > 1 meant until round 6 and still means under CHORDVIS_TILE_NEXT=0 -- costs a 30 k-cluster frame 3 / 7 %.) */
int chordvis_set_tile_schedule_keep(ChordCtx* ctx, uint32_t frames);
uint32_t chordvis_tile_schedule_keep(ChordCtx* ctx);

/* allocateGBufferTextures (render_textures.cpp:20-45): size the visibility target.  deviceVisibility
 * may be a caller-owned device buffer of chordvis_visibility_words(ctx) uint64 (used for the
 * multi-GPU all-gather), or NULL for a context-owned one. */
int chordvis_allocate_gbuffer(ChordCtx* ctx, uint32_t width, uint32_t height, uint64_t* deviceVisibility);

int chordvis_set_limits(ChordCtx* ctx, const ChordLimits* limits);
/* Multi-GPU screen ownership (the reference is single-device): the screen is cut into the rasterizer's 64 x 64-pixel
 * tiles and every tile belongs to one rank -- a table tile -> owner, row-major over the tile grid (ceil(W / 64) columns), the
 * same on every rank of a frame.  The visibility buffer of a sharded context is stored rank-major and tile-linear: a rank's
 * tiles sit in consecutive slots of 64 x 64 words (64 rows of 64) of the rank's chunk, chunk = the largest tile count of the
 * current map in slots (ceil(tiles / ranks) under the default map, up to chordvis_tile_slot_capacity under a weighted one), so one
 * in-place all-gather reassembles the frame and a de-tile kernel restores row-major.  ranks == 1: plain row-major.
 * chordvis_set_shard installs the default map, chordvis_tile_layout(width, height, ranks, NULL, ...): the tile grid is walked
 * along a generalised Hilbert curve and cut into `ranks` runs of equal length -- compact regions, so that few clusters touch
 * two ranks' tiles (such a cluster is set up by both). */
int chordvis_set_shard(ChordCtx* ctx, uint32_t ranks, uint32_t rank);
uint32_t chordvis_tile_count(uint32_t width, uint32_t height);
uint32_t chordvis_tile_slots_per_rank(uint32_t width, uint32_t height, uint32_t ranks);   /* ceil(tiles / ranks): a rank's tiles under the default map */
/* Tile slots a sharded context ALLOCATES per rank: a quarter more than that, room for a load-balanced map to give a rank of
 * light tiles more of them.  What the all-gathers move is ranks x (the largest tile count of the current map) slots
 * (chordvis_visibility_chunk_words and the exchange chunk sizes follow the map). */
uint32_t chordvis_tile_slot_capacity(uint32_t width, uint32_t height, uint32_t ranks);
/* The map as a function (host only, deterministic: every rank computes the same table from the same inputs).  loads NULL:
 * the default above.  loads = bin entries per tile of a rendered frame (chordvis_read_tile_loads): regions of equal LOAD along
 * the same curve; tiles heavier than a quarter of a rank's share are placed one by one (heaviest first, least loaded rank),
 * near-empty tiles fill every rank up to its tile count, each joining the region next to it on the curve while that region has
 * room.  Never more than maxTilesPerRank tiles per rank (0: ceil(tiles / ranks);
 * chordvis_rebalance passes chordvis_tile_slot_capacity). */
int chordvis_tile_layout(uint32_t width, uint32_t height, uint32_t ranks, const uint32_t* loads, uint32_t maxTilesPerRank, uint8_t* ownersOut);
/* An explicit map, between frames (drains frames in flight; the history HZB carries over: it is not sharded).  owners NULL:
 * back to the default map. */
int chordvis_set_tile_owners(ChordCtx* ctx, const uint8_t* owners, uint32_t tiles);
int chordvis_get_tile_owners(ChordCtx* ctx, uint8_t* ownersOut, uint32_t tiles);
/* Bin entries per tile of the last frame, EVERY rank's tiles (they travel in the end-of-frame exchange); waits for the stream. */
int chordvis_read_tile_loads(ChordCtx* ctx, uint32_t* loadsOut, uint32_t tiles);
/* read_tile_loads -> tile_layout -> set_tile_owners.  Every rank calls it at the same frame boundary (same loads -> same map).
 * imbalancePermille (may be NULL): the OLD map's heaviest rank / the mean, x 1000. */
int chordvis_rebalance(ChordCtx* ctx, uint32_t* imbalancePermille);
/* Number of uint64 words of the (rank-major, tile-linear when sharded) visibility buffer as allocated, and of one rank's chunk
 * under the current tile map (the all-gather's count; rank r's chunk starts at r x that). */
uint64_t chordvis_visibility_words(ChordCtx* ctx);
uint64_t chordvis_visibility_chunk_words(ChordCtx* ctx);
/* Device pointer of the visibility buffer in use (row-major when ranks == 1). */
uint64_t* chordvis_visibility_ptr(ChordCtx* ctx);

/* ------------------------------------------------------------------ passes (stream-ordered) */

/* addClearGbufferPass — render_textures.cpp:74-102 (visibility = 0, depth = 0.0). */
int chordvis_clear_gbuffer(ChordCtx* ctx);

/* instanceCulling — instance_culling.cpp:83-161 (instanceCullingCS + clusterGroupCullingCS).
 * Slots are assigned in deterministic (objectId, groupIdx, meshlet) order. */
int chordvis_instance_culling(ChordCtx* ctx, ChordCountAndCmd* out);

/* detail::hzbCulling — instance_culling.cpp:286-351 (hzbMainViewCullingCS).
 * bFirstStage: project with last-frame matrices, also emit the rejected list. */
int chordvis_hzb_culling(ChordCtx* ctx, const ChordHZB* hzb, int bFirstStage, ChordCountAndCmd in,
                         ChordCountAndCmd* outVisible, ChordCountAndCmd* outRejected);

/* renderMesh — mesh_raster.cpp:208-254: the 4 material buckets (filterPipeForVisibility + renderMeshRasterPipe, alphaMode
 * 0/1 x two-sided 0/1) collapse into one software-raster pass that reads bTwoSided and alphaMode per cluster.  Masked
 * materials (mesh_raster.hlsl:34-38,107-112,198-204) are alpha-tested per pixel with a software texture fetch whose
 * level of detail and filter are pinned (the reference leaves them to the sampler hardware);
 * blended materials (alphaMode 2) are in no bucket and draw nothing, as in the reference. */
int chordvis_render_mesh(ChordCtx* ctx, ChordCountAndCmd in);

/* gltfVisibilityRenderingStage0 — mesh_raster.cpp:269-311.  hzbPrev NULL/invalid or HZB culling
 * disabled => draws `in`, *shouldStage1 = 0. */
int chordvis_visibility_stage0(ChordCtx* ctx, const ChordHZB* hzbPrev, ChordCountAndCmd in,
                               ChordCountAndCmd* outRejected, int* shouldStage1);

/* gltfVisibilityRenderingStage1 — mesh_raster.cpp:313-329. */
int chordvis_visibility_stage1(ChordCtx* ctx, const ChordHZB* hzb, ChordCountAndCmd in);

/* buildHZB — hzb.cpp:38-227 from the depth half of the visibility words.
 * slot 0 = temporary (post stage 0), slots 1/2 = history ping-pong. */
int chordvis_build_hzb(ChordCtx* ctx, int bBuildMin, int bBuildMax, int bBuildValidRange, int slot, ChordHZB* out);

/* DeferredRenderer::render hot segment (renderer.cpp:315-345,489): clear -> instanceCulling ->
 * stage0 -> [buildHZB -> stage1] -> buildHZB(min,max,validRange); keeps the HZB as history for
 * the next call.  On a sharded context (ranks > 1) this needs a communicator (chordvis_comm_init_rank below) and runs the
 * three phases with the two all-gathers in between; a host that owns its collectives (e.g. torch.distributed) drives
 * the phases itself -- on the context's stream, so that kernels and collectives are ordered:
 *   chordvis_frame_phase_cull  (optional; 2..8 ranks, flat cull mode) the sharded group cull (SURVEY 8e "shard by object range"): the
 *                           object pass + the group / meshlet tests of THIS rank's share of the group instances only, one word per group
 *                           instance into the rank's chunk of the cull exchange buffer: byte i = the set of ranks whose tiles meshlet i of the
 *                           group touches, 0 = culled.  A frame that starts at phase a instead tests every group on every rank.
 *   [all-gather the cull exchange buffer: chordvis_cull_exchange_ptr, chunk = chordvis_cull_exchange_chunk_bytes]
 *   chordvis_frame_phase_a  clear .. stage 0 raster of the rank's tiles; the tile kernel also reduces every tile to its HZB texels
 *                           (mips 0..5), into the tile's slots of the two exchange buffers
 *   [all-gather the mid-frame exchange buffer: chordvis_hzb_exchange_ptr, chunk = chordvis_hzb_exchange_chunk_halves]
 *   chordvis_frame_phase_b  HZB chain from the exchanged texels, stage 1 cull + raster
 *   [all-gather the end-of-frame exchange buffer (chordvis_hzb_final_exchange_ptr / _chunk_bytes) and the visibility buffer, in place]
 *   chordvis_frame_phase_c  row-major copy of the image, history HZB from the exchanged texels, history swap            */
int chordvis_render_frame(ChordCtx* ctx);
int chordvis_frame_phase_cull(ChordCtx* ctx);
/* cull exchange buffer: ranks x (B x 256 mask words + B triangle sums), B = ceil(count blocks / ranks); made on the first request after
 * chordvis_upload_scene + chordvis_set_shard (NULL / 0 when the sharded cull does not apply: one rank, more than 8, no scene) */
uint32_t* chordvis_cull_exchange_ptr(ChordCtx* ctx);
uint64_t chordvis_cull_exchange_chunk_bytes(ChordCtx* ctx);   /* one rank */
int chordvis_frame_phase_a(ChordCtx* ctx);
int chordvis_frame_phase_b(ChordCtx* ctx);
int chordvis_frame_phase_c(ChordCtx* ctx);
/* Pipelined form of phase c, for hosts that let the visibility all-gather of frame i travel beside frame i + 1 (two frames'
 * words alive: chordvis_swap_visibility before every phase a).  The history HZB needs only the small end-of-frame exchange:
 *   [all-gather chordvis_hzb_final_exchange_ptr]
 *   chordvis_frame_phase_c_finish   the history chain from the exchanged texels, history swap; the frame is over
 *   [whenever the visibility all-gather of the frame has landed: chordvis_frame_resolve_visibility, on any stream]   */
int chordvis_frame_phase_c_finish(ChordCtx* ctx);
int chordvis_frame_resolve_visibility(ChordCtx* ctx, void* hipStream /* NULL: the context's */);
int chordvis_swap_visibility(ChordCtx* ctx);
/* end-of-frame exchange buffer: per tile slot the min and max chains' texels (mips 0..5), the tile's valid range and bin length */
uint16_t* chordvis_hzb_final_exchange_ptr(ChordCtx* ctx);
uint64_t chordvis_hzb_final_exchange_chunk_bytes(ChordCtx* ctx);   /* one rank */
int chordvis_reset_history(ChordCtx* ctx);
/* mid-frame exchange buffer: per tile slot the min chain's texels (mips 0..5, f16) after the first raster pass, rank-major */
uint16_t* chordvis_hzb_exchange_ptr(ChordCtx* ctx);
uint64_t chordvis_hzb_exchange_halves(ChordCtx* ctx);        /* whole buffer */
uint64_t chordvis_hzb_exchange_chunk_halves(ChordCtx* ctx);  /* one rank     */
/* row-major visibility after phase_c (== chordvis_visibility_ptr when ranks == 1) */
uint64_t* chordvis_resolved_visibility_ptr(ChordCtx* ctx);

/* ------------------------------------------------------------------ multi-GPU (the reference is single-device,
 * graphics.cpp:524-548).  The frame shards by screen tiles (chordvis_set_shard); the exchanges of a sharded frame -- the owned
 * tiles' HZB texels between the raster passes, their HZB texels and visibility words at the end -- are issued by the library,
 * so the host keeps ONE call per frame, like DeferredRenderer::render (renderer.cpp:319-345). */

/* (a) one process per GPU (torch.distributed / MPI hosts): attach an RCCL communicator to a sharded context; from then on
 * chordvis_render_frame(ctx) runs phase a -> ncclAllGather -> phase b -> ncclAllGather -> phase c on the context's stream.
 * Rank 0 makes the id (ncclGetUniqueId), the host distributes the 128 bytes by its own means, every rank calls init_rank
 * after chordvis_set_shard(nranks, rank).  librccl.so is resolved at run time, preferring a copy already loaded
 * into the process (CHORDVIS_RCCL=<path> overrides). */
#define CHORDVIS_UNIQUE_ID_BYTES 128
int chordvis_comm_unique_id(void* out128);
int chordvis_comm_init_rank(ChordCtx* ctx, uint32_t nranks, uint32_t rank, const void* id128);
int chordvis_comm_destroy(ChordCtx* ctx);
/* Pipelined frames over RCCL (the ChordGroup form: chordvis_group_set_pipelined below).  id128: a SECOND unique id (the same
 * on every rank) for the communicator that carries the image of frame i, on a stream of its own, beside frame i + 1; the
 * history HZB then waits only for the small end-of-frame exchange, not for the image.  chordvis_readback_visibility / chordvis_visibility_mark / chordvis_wait_visibility wait for the image.
 * NULL switches back to the plain protocol.  The context must own its visibility buffer.
 * Two communicators of one process run on two streams of one device, and RCCL orders collectives per communicator only.
 * What makes that safe here: (1) each communicator is used from exactly one stream, always the same one; (2) every rank
 * issues the same collectives in the same order on each communicator (frame after frame: small, image, small -- whether the
 * mid-frame exchange happens is decided from state every rank shares); (3) no collective of one communicator waits, on the
 * device, for a LATER collective of the other on any rank: the image gather waits for the compute stream's "phase b done"
 * event, which lies before the next small exchange.  A host that adds collectives of its own must keep (1)-(3), and must not
 * run a blocking call (hipMalloc, hipFree, a synchronous copy) on one rank between the two enqueues while its peers are
 * already inside them -- the usual NCCL multi-communicator rule.  With one rank (all a one-GPU box can host) the call sets up
 * the same second communicator, stream and events; the frame is the unsharded one followed by the image step.
 * A frame that fails on one rank (capacity, HIP error) still issues every collective of the frame, so its peers do not hang,
 * returns the first error, and leaves the context outside a frame: the next chordvis_render_frame starts clean (its history is
 * the last complete frame's only if the host calls chordvis_reset_history on EVERY rank; otherwise that rank culls against a
 * chain its peers do not share and the images may differ -- treat a failed frame as a reason to reset or to stop). */
int chordvis_comm_set_pipelined(ChordCtx* ctx, const void* id128);
/* NCCL_VERSION_CODE of the loaded library, ranks of ctx's communicator (0 = none; ctx may be NULL), where librccl came from */
int chordvis_comm_info(ChordCtx* ctx, int* ncclVersion, uint32_t* nranks, char* libraryOrigin, uint32_t originBytes);

/* (b) one process, n devices: one context + one host thread per device; the exchanges are direct all-gathers -- every rank
 * pushes its chunk to each peer with its own hipMemcpyPeerAsync (n-1 concurrent copies per rank, one per xGMI link).
 * Ordinals may repeat (protocol tests on a one-GPU box).  The scene is replicated; per-rank readback and stats go through
 * chordvis_group_ctx(group, rank) (borrowed; every rank ends a frame with the complete row-major image and HZB). */
typedef struct ChordGroup ChordGroup;
int chordvis_create_group(uint32_t n, const int* deviceOrdinals, ChordGroup** outGroup);
int chordvis_destroy_group(ChordGroup* group);
uint32_t chordvis_group_size(ChordGroup* group);
ChordCtx* chordvis_group_ctx(ChordGroup* group, uint32_t rank);
const char* chordvis_group_last_error(ChordGroup* group);
int chordvis_group_set_limits(ChordGroup* group, const ChordLimits* limits);
int chordvis_group_upload_scene(ChordGroup* group, const ChordSceneDesc* scene);
int chordvis_group_allocate_gbuffer(ChordGroup* group, uint32_t width, uint32_t height);
/* chordvis_rebalance on every rank (drains the frames in flight first) */
int chordvis_group_rebalance(ChordGroup* group, uint32_t* imbalancePermille);
int chordvis_group_update_objects(ChordGroup* group, const ChordObject* hostObjects, uint32_t count);
int chordvis_group_set_view(ChordGroup* group, const ChordCameraView* view, const ChordInstanceCullingView* instanceView, uint32_t switchFlags);
/* DeferredRenderer::render hot segment (renderer.cpp:315-345,489) on n devices; returns when the frame is ENQUEUED on every
 * device (no host synchronisation; chordvis_group_sync waits) */
int chordvis_group_render_frame(ChordGroup* group);
int chordvis_group_sync(ChordGroup* group);
/* mean host time per frame (ms) each rank's worker spent inside chordvis_group_render_frame since the last call; n = group size */
int chordvis_group_enqueue_ms(ChordGroup* group, double* msPerRank, uint32_t n);
/* Pipelined frames: chordvis_group_render_frame returns once frame i is enqueued with its visibility all-gather and row-major
 * copy running beside whatever follows (frame i + 1); every rank's history HZB is complete at the end of the call's work as
 * before.  chordvis_readback_visibility / the consumer entry points of a rank's context wait for ITS image;
 * chordvis_readback_previous_visibility reads the frame before the last submitted one. */
int chordvis_group_set_pipelined(ChordGroup* group, int enable);

/* ------------------------------------------------------------------ depth-only views (what renderShadow runs per
 * cascade, mesh_raster.cpp:331-546).  Every pass of the reference takes (instanceViewId, instanceViewOffset) -- a buffer of
 * InstanceCullingViewInfo and an index into it (gltf_rendering.h:38-43,54-64,89-110); here the buffer is set once and the
 * passes take the offset.  Views have their own square size (CascadeShadowMapConfig::cascadeDim, render_helper.h:469). */
typedef struct ChordDepthTarget {     /* a VK_FORMAT_D32_SFLOAT image (mesh_raster.cpp:407-416): device floats, row-major, reverse-Z, clear 0 */
    float*   depth;
    uint32_t width, height;
} ChordDepthTarget;
/* after upload_scene: targets + work lists for `viewCount` views of dim x dim pixels */
int chordvis_allocate_depth_views(ChordCtx* ctx, uint32_t dim, uint32_t viewCount);
/* the InstanceCullingViewInfo[] of the views ("CascadeViewInfos", mesh_raster.cpp:417-421); host array, copied */
int chordvis_set_instance_views(ChordCtx* ctx, const ChordInstanceCullingView* hostViews, uint32_t count);
/* instanceCulling(queue, ctx, instanceCullingViewInfo, instanceCullingViewInfoOffset) for a view other than the main one
 * (mesh_raster.cpp:452): object / meshlet frustum tests against the view (orthoFrustumCulling for an orthographic one,
 * base.hlsli:251-272), LOD cut by the MAIN camera (chordvis_set_view).  The list stays valid until the next call. */
int chordvis_instance_culling_view(ChordCtx* ctx, uint32_t instanceViewOffset, ChordCountAndCmd* out);
/* detail::hzbCullingGeneric (instance_culling.cpp:232-284, hzb_culling_generic.hlsl): one-pass occlusion cull of a view's list
 * against an HZB of that view's depth (chordvis_build_hzb_from_depth) */
int chordvis_hzb_culling_generic(ChordCtx* ctx, const ChordHZB* hzb, float extentScale, uint32_t instanceViewOffset,
                                 int bObjectUseLastFrameProject, ChordCountAndCmd in, ChordCountAndCmd* out);
/* queue.clearDepthStencil(depth, 0) + renderMeshDepth(PASS_TYPE_DEPTH) (mesh_raster.cpp:159-206,500-522): both alpha buckets,
 * cull mode NONE, depth test GREATER_OR_EQUAL, optional depth clamp and vkCmdSetDepthBias(const, 0, slope).  The view must
 * be the one of the last chordvis_instance_culling_view. */
int chordvis_render_mesh_depth(ChordCtx* ctx, uint32_t instanceViewOffset, int bDepthClamped, float depthBiasConst, float depthBiasSlope,
                               ChordCountAndCmd in, ChordDepthTarget* out);
/* buildHZB(queue, depth, true, false, false) on a depth target (mesh_raster.cpp:466,527) */
int chordvis_build_hzb_from_depth(ChordCtx* ctx, const ChordDepthTarget* depth, ChordHZB* out);
int chordvis_readback_depth(ChordCtx* ctx, const ChordDepthTarget* depth, float* host);
/* renderShadow (mesh_raster.cpp:331-546): cascade setup, then per cascade from the farthest to the nearest: instanceCulling for
 * its view, hzbCullingGeneric against the previous cascade's HZB (or, for the first one, against the HZB of its own cached
 * depth from last frame's view), clear + renderMeshDepth with depth clamp and the configured bias, buildHZB.  The context
 * keeps the depth images and views from call to call (CascadeShadowHistory): with an unchanged config and light direction
 * the far cascades are refreshed one per tick (isCascadeCacheValid, cascade_setup.hlsl:8-22).  validDepthMinMax: the
 * valid range of the main view's last buildHZB (chordvis_readback_hzb) or NULL.  outRenderedMask: bit i = cascade i was
 * re-rendered by this call. */
int chordvis_render_shadow(ChordCtx* ctx, const ChordCascadeConfig* config, const float lightDir[3], const uint32_t validDepthMinMax[2],
                           uint32_t tickCount, int bHzbCulling, ChordDepthTarget* outDepths /* [cascadeCount] or NULL */,
                           ChordInstanceCullingView* outViews /* [cascadeCount] or NULL */, uint32_t* outRenderedMask);
/* chordvis_stats of the depth views' last pass (overflow flag, record / bin counts) */
int chordvis_depth_view_stats(ChordCtx* ctx, ChordStats* out);

/* handles of the last frame (post-instanceCulling list: consumer contract, lighting.hlsl:318-345) */
int chordvis_last_frame_cmds(ChordCtx* ctx, ChordCountAndCmd* out);
int chordvis_history_hzb(ChordCtx* ctx, ChordHZB* out);

/* ------------------------------------------------------------------ consumers' first step */
/* visibilityMark — visibility_tile.cpp:20-57 (tilerMarkerCS, visibility_tile.hlsl:65-134): marks, per 8x8 pixels of
 * the (resolved, row-major) visibility buffer, which material shading types occur.  drawedMeshletCmd = the list
 * the visibility ids index, i.e. chordvis_last_frame_cmds (renderer.cpp:354,359). */
int chordvis_visibility_mark(ChordCtx* ctx, ChordCountAndCmd drawedMeshletCmd, ChordTileMarker* out);
/* Consumers that read chordvis_resolved_visibility_ptr() on a stream of their own (lighting.hlsl:318-329 is the reference's):
 * orders `hipStream` (NULL: the context's) behind the completion of the last submitted frame's image.  After pipelined
 * ChordGroup frames the image is gathered beside the context's stream, so this -- or one of the library's own consumers /
 * read-backs, which wait by themselves -- must come before the first read. */
int chordvis_wait_visibility(ChordCtx* ctx, void* hipStream);
/* prepareShadingTileParam — visibility_tile.cpp:59-110 (tilePrepareCS + prepareTileParamCS, visibility_tile.hlsl:136-219) */
int chordvis_prepare_shading_tile_param(ChordCtx* ctx, uint32_t shadingType, const ChordTileMarker* marker, ChordShadingTiles* out);

/* ------------------------------------------------------------------ readback / stats (synchronize) */

int chordvis_readback_visibility(ChordCtx* ctx, uint64_t* hostWords /* width*height, row-major */);
int chordvis_readback_previous_visibility(ChordCtx* ctx, uint64_t* hostWords);   /* pipelined sharded frames: the other buffer pair */
int chordvis_readback_cmds(ChordCtx* ctx, ChordCountAndCmd handle, ChordDrawCmd* hostCmds, uint32_t cap, uint32_t* outCount);
int chordvis_readback_hzb(ChordCtx* ctx, const ChordHZB* hzb, uint16_t* hostMin, uint16_t* hostMax, uint32_t hostValidRange[2]);
/* host: 4 words per marker texel; tiles: 2 words per tile (at most hostCapacity tiles are copied) */
int chordvis_readback_tile_marker(ChordCtx* ctx, const ChordTileMarker* marker, uint32_t* host);
int chordvis_readback_shading_tiles(ChordCtx* ctx, const ChordShadingTiles* tiles, uint32_t* hostTiles, uint32_t hostCapacity,
                                    uint32_t* hostCount, uint32_t hostDispatchArgs[4]);
/* upload an HZB min chain from the host (tests: feed a known history) into history */
int chordvis_upload_history_hzb(ChordCtx* ctx, const uint16_t* hostMin);

/* mode (low 8 bits) 0: off.  1: GPU timestamps of the last frame.  2: accumulate over frames until the
 * next chordvis_stats, which then reports per-frame averages (and restarts the accumulation).
 * mode >> 8 = sampling period P (0/1 = every frame): only every P-th frame is stamped, because each
 * hipEventRecord between two kernels costs ~5 us of stream idle time on MI355X. */
int chordvis_enable_timers(ChordCtx* ctx, int mode);
int chordvis_stats(ChordCtx* ctx, ChordStats* out);
/* Measurement-only switches of the raster kernels (kernels_raster.hip DBG_*).  The clocks (16, 512) and the ablation switches
 * are compiled into the kernels only with -DRASTER_PROFILE=1 / -DRASTER_ABLATION=1 (python chord_amd/build.py --tag NAME -D...;
 * load it with CHORDVIS_LIB): the product library REFUSES them (CHORDVIS_E_INVALID), because testing them at run time costs the
 * register-bound kernels 3-5 %.  1 no pixel writes, 2 setup
 * emits no records / bins, 16 per-tile clocks (chordvis_debug_tile_profile), 32 skip the per-lane scan of tiny
 * triangles, 64 tile kernel of later passes returns at once, 128 no tile-out, 256 tile-out without the HZB
 * reduction, 512 setup-kernel phase clocks (chordvis_debug_setup_profile), 1024 fused tile-out skips the visibility
 * stores, 2048 never split long bins, 4096 / 8192 / 16384 tile kernel skips its row units / entry set-up / the bin
 * altogether.  0 = production; any of those voids parity.  These switches do NOT change results (tests run them): 32768 small
 * clusters never leave the setup kernel as pixel blocks, 65536 every launch takes the setup kernel's pixel-block body
 * (by default it does when a launch has more than one cluster per 16 pixels), 131072 chordvis_render_frame launches
 * hzb_tail_kernel between the raster passes instead of letting the phase-1 cull reduce HZB levels 6.. itself, 262144 the
 * pixel-block kernel always runs its hot-tile variant and treats a bin of 64 entries as hot (by default the variant is chosen
 * when the previous frame had a bin of 65536 entries or more), 524288 the library's own sharded frames (chordvis_comm_*, ChordGroup)
 * keep the replicated group cull instead of the sharded one (A / B measurements; every rank of a frame must agree). */
int chordvis_set_debug(ChordCtx* ctx, uint32_t flags);
/* measurement / test aid: fills EVERY rank's chunk of the cull exchange buffer on this context for the current view (what the
 * all-gather would deliver), so that one rank of a sharded frame can be timed alone on one device (tools/shard_time.py) */
int chordvis_debug_fill_cull_exchange(ChordCtx* ctx);
/* debugging aid: raw read of an internal buffer (0 tile counts, 1 fixed bins, 2 chunk table, 3 bin pool, 4 / 5 32- / 48-byte records) */
int chordvis_debug_read(ChordCtx* ctx, int which, uint64_t offset, uint64_t bytes, void* host);
/* debugging aid: non-zero words in the split-tile accumulation slabs (must be 0 between raster passes) */
int chordvis_debug_slab_nonzero(ChordCtx* ctx, uint64_t* count);
/* measurement aid: ms per frame of 2*pairs stream-launched frames vs the same frames replayed from a hipGraph */
int chordvis_debug_graph_frames(ChordCtx* ctx, uint32_t pairs, float* msPerFrameStream, float* msPerFrameGraph);
/* debug bit 512: per-wave phase ticks (10 ns) of the setup kernel summed over waves: header wait / vertex / triangle / reserve / emit.
 * The clocks of bits 16 and 512 are compiled in only with -DRASTER_PROFILE=1 (python chord_amd/build.py --tag prof -DRASTER_PROFILE=1:
 * their accumulators cost the product kernels scalar registers they do not have); the product library leaves the ticks zero. */
int chordvis_debug_setup_profile(ChordCtx* ctx, int pass, uint64_t hostTicks[5], uint32_t* waves);
int chordvis_debug_tile_profile(ChordCtx* ctx, int pass, uint64_t* hostTicks, uint32_t* hostCounts, uint32_t capacity);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* CHORDVIS_H */
