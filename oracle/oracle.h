/*
 * oracle.h — CPU restatement of chord's visibility hot path.  TEST INFRASTRUCTURE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (chord_amd/, libchordvis.so) never does.
 *
 * PARITY UNPINNED: the reference has no golden vectors, known-answer tests or
 * fixtures for this path (application/unit_test/main.cpp:6-26 tests only the
 * job system) and cannot be built or run here (MSVC + Vulkan mesh shaders +
 * DXC; CMakeLists.txt:94-96, source/shader_compiler/compiler.cpp:7-13).  This
 * restatement therefore pins the implementation-defined choices the reference
 * leaves to the driver / fixed-function hardware (see "Canonical arithmetic"
 * in oracle.c) and is itself pinned by analytic known-answer tests under
 * tests/ (raster rules, HZB brute force, LOD cut, frustum KATs).
 *
 * Each function cites the reference file:line whose algorithm it follows.
 */
#ifndef CHORD_ORACLE_H
#define CHORD_ORACLE_H

#include "../include/chordvis_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OrcRasterStats {
    uint64_t clusters;           /* draw commands rastered                                  */
    uint64_t trianglesSubmitted; /* sum of meshlet triangle counts (the Gtri/s unit)        */
    uint64_t trianglesBackface;  /* culled by mesh_raster.hlsl:143-149                      */
    uint64_t trianglesNear;      /* :152-155                                                */
    uint64_t trianglesOffscreen; /* :168-171                                                */
    uint64_t trianglesSmall;     /* :174-179                                                */
    uint64_t trianglesClipped;   /* survivors that needed the homogeneous clipper           */
    uint64_t trianglesRastered;  /* survivors scan-converted (after snapped-area rejection) */
    uint64_t fragments;          /* covered pixel centres (before the depth test)           */
    uint64_t fragmentsClipped;   /* covered pixel centres a masked material's clip() dropped (not in `fragments`) */
} OrcRasterStats;

/* Screen ownership for the multi-GPU shard: the screen is cut into 64 x 64-pixel tiles, tile (tx, ty) belongs to rank
 * owners[ty * tilesX + tx]; a rank writes only the pixels of its tiles.  NULL / ranks == 1 => all. */
typedef struct OrcShard {
    const uint8_t* owners;
    uint32_t tilesX;
    uint32_t ranks;
    uint32_t rank;
} OrcShard;

uint16_t orc_f32_to_f16(float f);      /* round-to-nearest-even */
float    orc_f16_to_f32(uint16_t h);

/* hzb.cpp:49-63 */
void orc_hzb_desc(uint32_t srcW, uint32_t srcH, ChordHZBDesc* out);
/* valid (sampled) extent of mip l: texels x <= ((W-1)>>1)>>l */
uint32_t orc_hzb_valid_w(const ChordHZBDesc* d, uint32_t l);
uint32_t orc_hzb_valid_h(const ChordHZBDesc* d, uint32_t l);

/* instance_culling.hlsl:47-131 (object stage only): visible[o] = 0/1 */
void orc_object_cull(const ChordSceneDesc* scene, const ChordInstanceCullingView* iv,
                     uint32_t flags, uint8_t* visible);

/* nanite_shared.hlsli:15-49 */
int orc_group_visible(const ChordCameraView* view, const ChordObject* obj, const ChordMeshletGroup* g);

/* nanite_shared.hlsli:51-91 */
int orc_meshlet_visible(uint32_t flags, const ChordInstanceCullingView* iv, const ChordObject* obj,
                        const ChordMeshlet* m, const ChordMaterial* mat);

/* instance_culling.hlsl:47-208 — both stages, canonical (objectId, group, meshlet) order.
 * Returns the number of commands (may exceed cap; only cap are written). */
uint32_t orc_instance_culling(const ChordSceneDesc* scene, const ChordCameraView* view,
                              const ChordInstanceCullingView* iv, uint32_t flags,
                              ChordDrawCmd* outCmds, uint32_t cap);

/* hzb_mainview_culling.hlsl:35-213.  phase 0 => last-frame matrices, emits
 * visible + rejected; phase 1 => current matrices, emits visible only
 * (outRejected may be NULL). Order of the input list is preserved. */
void orc_hzb_culling(const ChordSceneDesc* scene, const ChordCameraView* view, uint32_t flags, int phase,
                     const ChordHZBDesc* hzb, const uint16_t* hzbMin,
                     const ChordDrawCmd* inCmds, uint32_t inCount,
                     ChordDrawCmd* outVisible, uint32_t* outVisibleCount,
                     ChordDrawCmd* outRejected, uint32_t* outRejectedCount);

/* hzb_one.hlsl:126-372 / hzb.hlsl:127-389 — depth = high 32 bits of the
 * visibility words.  hzbMax and validRange may be NULL. */
void orc_hzb_build(const uint64_t* vis, uint32_t W, uint32_t H, const ChordHZBDesc* desc,
                   uint16_t* hzbMin, uint16_t* hzbMax, uint32_t validRange[2]);

/* mesh_raster.hlsl:51-210 + the fixed-function state it drives, canonical
 * software scan conversion into the packed 64-bit visibility words. */
void orc_raster(const ChordSceneDesc* scene, const ChordInstanceCullingView* iv,
                const ChordDrawCmd* cmds, uint32_t count, const OrcShard* shard,
                uint64_t* vis, OrcRasterStats* stats);

/* Scan-convert one screen-space triangle given snapped 24.8 coordinates
 * (exposed for the raster-rule known-answer tests). */
/* renderMeshDepth (mesh_raster.cpp:159-206): PASS_TYPE_DEPTH of `cmds` for view `iv` into `vis` (NOT cleared here); the D32
 * image is the high word of every element, the low word stays 0 */
void orc_raster_depth(const ChordSceneDesc* scene, const ChordInstanceCullingView* iv, const ChordDrawCmd* cmds, uint32_t count,
                      int depthClamp, float biasConst, float biasSlope, uint64_t* vis, OrcRasterStats* stats);
/* hzb_culling_generic.hlsl:37-172; returns the number of commands kept in outCmds (input order) */
uint32_t orc_hzb_culling_generic(const ChordSceneDesc* scene, const ChordInstanceCullingView* iv, const double mainCameraWorldPos[3],
                                 uint32_t flags, float extentScale, int bObjectUseLastFrameProject,
                                 const ChordHZBDesc* hzb, const uint16_t* hzbMin,
                                 const ChordDrawCmd* inCmds, uint32_t inCount, ChordDrawCmd* outCmds);

/* canonical masked-material texture fetch (oracle.c header item 9): alpha of texture `tex` (NULL: 1) at (u, v) */
float orc_sample_alpha(const ChordTexture* tex, const ChordSampler* smp, uint32_t level, int linear, float u, float v);
/* level / filter a triangle with doubled snapped area absArea2 and texture coordinates u[], v[] samples `mat`'s texture at */
uint32_t orc_mask_level(const ChordSceneDesc* scene, const ChordMaterial* mat, int64_t absArea2, const float u[3], const float v[3], int* linear);

void orc_raster_snapped_triangle(const int32_t X[3], const int32_t Y[3], const float d[3],
                                 int twoSided, uint32_t payload, uint32_t W, uint32_t H,
                                 const OrcShard* shard, uint64_t* vis, OrcRasterStats* stats);

/* Frame logic — mesh_raster.cpp:269-329, renderer.cpp:319-345,489.
 * prevHzbMin == NULL => no history (first frame): whole list drawn, no stage 1.
 * outCmds receives the post-instanceCulling list (consumer contract,
 * lighting.hlsl:318-345). outHzbMin/outHzbMax/outValidRange = the HZB kept
 * for the next frame. counts[0..3] = {instanceCulled, stage0Visible,
 * stage0Rejected, stage1Visible}. */
void orc_frame(const ChordSceneDesc* scene, const ChordCameraView* view, const ChordInstanceCullingView* iv,
               uint32_t flags, const uint16_t* prevHzbMin, const OrcShard* shard,
               uint64_t* vis, ChordDrawCmd* outCmds, uint32_t cmdCap, uint32_t counts[4],
               uint16_t* outHzbMin, uint16_t* outHzbMax, uint32_t outValidRange[2],
               OrcRasterStats* stats);

/* All-cores replay of orc_frame (SURVEY 8d CPU baseline): instance culling over object ranges, HZB culls over command
 * ranges, clusters rastered by `threads` threads into per-thread tile-private images merged by max, HZB levels over row
 * ranges.  Identical outputs to orc_frame (unsharded). */
void orc_frame_mt(const ChordSceneDesc* scene, const ChordCameraView* view, const ChordInstanceCullingView* iv,
                  uint32_t flags, const uint16_t* prevHzbMin, uint32_t threads,
                  uint64_t* vis, ChordDrawCmd* outCmds, uint32_t cmdCap, uint32_t counts[4],
                  uint16_t* outHzbMin, uint16_t* outHzbMax, uint32_t outValidRange[2],
                  OrcRasterStats* stats);

/* One raster leg on `threads` threads that share the image (compare-and-swap per fragment): the round-2 form, kept for
 * tests that raster a list of their own. */
void orc_raster_mt(const ChordSceneDesc* scene, const ChordInstanceCullingView* iv,
                   const ChordDrawCmd* cmds, uint32_t count, uint32_t threads,
                   uint64_t* vis, OrcRasterStats* stats);

/* Visibility tile marker — visibility_tile.hlsl:39-134 (tilerMarkerCS): one uint4 (128 shading-type bits)
 * per 8x8 pixels; marker holds 4 words per texel, markerDim = ceil(dim / 8) (visibility_tile.cpp:31).
 * cmds = the post-instanceCulling list the visibility ids index (renderer.cpp:354,359). */
void orc_visibility_mark(const ChordSceneDesc* scene, const uint64_t* vis, uint32_t W, uint32_t H,
                         const ChordDrawCmd* cmds, uint32_t cmdCount, uint32_t* marker);

/* Shading tile list — visibility_tile.hlsl:136-219 (tilePrepareCS, prepareTileParamCS): pixel origins of
 * the 8x8 tiles whose marker has the bit of shadingType, their count and the indirect dispatch argument
 * {(count+3)/4, 1, 1, 1}.  The reference's list order is scheduling-dependent (one atomic per wave); this
 * restatement emits in workgroup / lane / sample order.  Returns the count. */
uint32_t orc_shading_tiles(const uint32_t* marker, uint32_t markerW, uint32_t markerH, uint32_t shadingType,
                           uint32_t* tiles, uint32_t dispatchArgs[4]);

#ifdef __cplusplus
}
#endif
#endif
