// Depth-only views: the passes renderShadow runs per cascade (mesh_raster.cpp:331-546) behind the C ABI.
//
//   chordvis_set_instance_views       <- the InstanceCullingViewInfo[] buffer (cascadeViewInfos; base.h:121-135)
//   chordvis_instance_culling_view    <- instanceCulling(queue, ctx, viewsSRV, cascadeId)       gltf_rendering.h:38-43, mesh_raster.cpp:452
//   chordvis_hzb_culling_generic      <- detail::hzbCullingGeneric                              instance_culling.cpp:232-284
//   chordvis_render_mesh_depth        <- clearDepthStencil + renderMeshDepth(PASS_TYPE_DEPTH)  mesh_raster.cpp:159-206,500-522
//   chordvis_build_hzb_from_depth     <- buildHZB(queue, depth, true, false, false)            mesh_raster.cpp:466,527
//
// A view of another size than the main render target needs its own tile bins, lists and visibility words; they live in
// a CHILD context of the cascade size that shares the parent's scene buffers (geometry, materials, textures, BVH) and
// stream.  Every kernel is the main path's: the object pass takes the view's translatedWorldToClip / frustum planes
// from the InstanceCullingViewInfo (orthographic for a cascade: base.hlsli:243-272) while the LOD cut keeps using the
// MAIN camera (instance_culling.hlsl:166-174), the rasterizer runs with cull mode NONE, no id output, depth clamp and
// depth bias, and the D32 image is the high half of the 64-bit words.

#include "device_layer.h"

#include <cstring>
#include <vector>

using namespace chord;

namespace {

int need_child(ChordCtx* c, const char* fn)
{
    if (!c) return CHORDVIS_E_INVALID;
    if (!c->depthCtx) {
        char buf[160];
        std::snprintf(buf, sizeof(buf), "%s: chordvis_allocate_depth_views must come first (after upload_scene)", fn);
        return fail(c, CHORDVIS_E_INVALID, buf);
    }
    return CHORDVIS_OK;
}

int child_fail(ChordCtx* c, int rc)
{
    if (rc && c->depthCtx) c->lastError = c->depthCtx->lastError;
    return rc;
}

void double3_of(const uint32_t words[8], double out[3]) { std::memcpy(out, words, sizeof(double) * 3); }   // GPUStorageDouble4

} // namespace

extern "C" {

int chordvis_allocate_depth_views(ChordCtx* c, uint32_t dim, uint32_t viewCount)
{
    if (!c || !c->sceneLoaded) return fail(c, CHORDVIS_E_INVALID, "allocate_depth_views: upload_scene must come first");
    if (dim < 64 || dim > 4096 || viewCount == 0 || viewCount > 32) return fail(c, CHORDVIS_E_INVALID, "allocate_depth_views: dim 64..4096, 1..32 views (kMaxCascadeCount)");
    if (c->shard.ranks > 1 || c->sharedScene) return fail(c, CHORDVIS_E_INVALID, "allocate_depth_views: not on a sharded or child context");
    if (c->depthCtx) { chordvis_destroy(c->depthCtx); c->depthCtx = nullptr; }
    for (float*& d : c->dDepthImages) if (d) { (void)hipFree(d); d = nullptr; }
    c->fusedDepthView = -1;                               // (the child's chain 0 goes with the child: no image of the new views is in it)
    ChordCtx* k = nullptr;
    int rc = chordvis_create(c->device, c->stream, &k);
    if (rc) return fail(c, rc, "allocate_depth_views: child context");
    if (!c->stream) { chordvis_destroy(k); return fail(c, CHORDVIS_E_INVALID, "allocate_depth_views: no stream"); }
    if (k->ownStream) {                                   // (parent stream is never null after create, but be safe)
        (void)hipStreamDestroy(k->stream); k->stream = c->stream; k->ownStream = false;
    }
    k->sharedScene = true;
    k->dPrims = c->dPrims; k->dGroups = c->dGroups; k->dMeshlets = c->dMeshlets; k->dGroupIndices = c->dGroupIndices;
    k->dMeshletData = c->dMeshletData; k->dPositions = c->dPositions; k->dObjStatic = c->dObjStatic; k->dGroupRefs = c->dGroupRefs;
    k->dMaterials = c->dMaterials; k->dTexAlpha = c->dTexAlpha; k->dTexcoords = c->dTexcoords; k->dBvhNodes = c->dBvhNodes;
    k->bvhComplete = c->bvhComplete; k->anyMasked = c->anyMasked; k->hPrims = c->hPrims; k->hObjStatic = c->hObjStatic;
    k->objectCount = c->objectCount; k->primCount = c->primCount; k->materialCount = c->materialCount; k->meshletCount = c->meshletCount;
    k->groupCount = c->groupCount; k->groupInstances = c->groupInstances; k->cmdCapacity = c->cmdCapacity; k->cullBlocks = c->cullBlocks;
    k->instTriangles = c->instTriangles; k->limitRecords = c->limitRecords; k->limitPoolChunks = c->limitPoolChunks; k->binMaxChunks = c->binMaxChunks;
    c->depthCtx = k;
    k->debugFlags = c->debugFlags;
    // a failure past this point leaves nothing half-made behind: the child, the images and depthDim go together
    auto undo = [&](int code) -> int {
        if (c->depthCtx) c->lastError = c->depthCtx->lastError;
        for (float*& d : c->dDepthImages) if (d) { (void)hipFree(d); d = nullptr; }
        c->dDepthImages.clear();
        c->depthDim = 0;
        c->fusedDepthView = -1;
        if (c->depthCtx) { chordvis_destroy(c->depthCtx); c->depthCtx = nullptr; }
        return code;
    };
    if ((rc = alloc_scene_work_buffers(k))) return undo(rc);
    k->dObjects = c->dObjects;
    k->sceneLoaded = true;
    if ((rc = chordvis_allocate_gbuffer(k, dim, dim, nullptr))) return undo(rc);
    c->dDepthImages.assign(viewCount, nullptr);
    for (uint32_t i = 0; i < viewCount; i++) {
        hipError_t e = hipMalloc((void**)&c->dDepthImages[i], sizeof(float) * (size_t)dim * dim);
        if (e == hipSuccess) e = hipMemsetAsync(c->dDepthImages[i], 0, sizeof(float) * (size_t)dim * dim, c->stream);
        if (e != hipSuccess) { const int code = fail(c, CHORDVIS_E_HIP, "allocate_depth_views: depth image", e); const std::string msg = c->lastError; undo(code); c->lastError = msg; return code; }
    }
    c->depthDim = dim;
    return CHORDVIS_OK;
}

int chordvis_set_instance_views(ChordCtx* c, const ChordInstanceCullingView* hostViews, uint32_t count)
{
    if (!c || !hostViews || count == 0 || count > 32) return fail(c, CHORDVIS_E_INVALID, "set_instance_views: 1..32 views");
    c->instanceViews.assign(hostViews, hostViews + count);
    return CHORDVIS_OK;
}

int chordvis_instance_culling_view(ChordCtx* c, uint32_t instanceViewOffset, ChordCountAndCmd* out)
{
    int rc = need_child(c, "instance_culling_view");
    if (rc) return rc;
    if (!c->viewSet) return fail(c, CHORDVIS_E_INVALID, "instance_culling_view: set_view (the main camera: the LOD cut uses it) must come first");
    if (instanceViewOffset >= c->instanceViews.size()) return fail(c, CHORDVIS_E_INVALID, "instance_culling_view: instanceViewOffset beyond set_instance_views");
    ChordCtx* k = c->depthCtx;
    const ChordInstanceCullingView& iv = c->instanceViews[instanceViewOffset];
    if ((uint32_t)iv.renderDimension[0] != k->width || (uint32_t)iv.renderDimension[1] != k->height)
        return fail(c, CHORDVIS_E_INVALID, "instance_culling_view: the view's renderDimension differs from the allocated depth views");
    k->dObjects = c->dObjects;
    k->cullMode = c->cullMode;
    k->hView.view = c->hView.view;                       // instance_culling.hlsl:166-174: LOD selection always by the main view
    k->hView.iv = iv;
    k->hView.flags = c->hView.flags;
    k->hView.width = k->width; k->hView.height = k->height;
    k->viewSet = true; k->viewDirty = true;
    k->depthViewCurrent = (int)instanceViewOffset;
    // a view's passes start here: ONE memset zeroes everything they count with (counters, list counts, both passes' tile bins),
    // instead of the seven small ones a raster pass outside a frame issues for itself (each a launch of its own)
    k->frameStateZeroBytes = offsetof(chord::FrameState, tileCount) + sizeof(uint32_t) * CHORD_TILECOUNT_STRIDE * ((size_t)2 * k->tilesX * k->tilesY);
    CHORD_HIP(c, hipMemsetAsync(k->dFrameState, 0, k->frameStateZeroBytes, k->stream));
    k->rasterCalls = 0; k->inFrame = true;
    launch_group_cull(k, k->lists[0]);
    CHORD_HIP(c, hipGetLastError());
    if (out) *out = k->lists[0].handle();
    return CHORDVIS_OK;
}

static int generic_cull(ChordCtx* c, const ChordHZB* hzb, float extentScale, const ChordInstanceCullingView& iv,
                        int bObjectUseLastFrameProject, ChordCountAndCmd in, ChordCountAndCmd* out);

int chordvis_hzb_culling_generic(ChordCtx* c, const ChordHZB* hzb, float extentScale, uint32_t instanceViewOffset,
                                 int bObjectUseLastFrameProject, ChordCountAndCmd in, ChordCountAndCmd* out)
{
    int rc = need_child(c, "hzb_culling_generic");
    if (rc) return rc;
    if (instanceViewOffset >= c->instanceViews.size()) return fail(c, CHORDVIS_E_INVALID, "hzb_culling_generic: instanceViewOffset beyond set_instance_views");
    return generic_cull(c, hzb, extentScale, c->instanceViews[instanceViewOffset], bObjectUseLastFrameProject, in, out);
}

static int generic_cull(ChordCtx* c, const ChordHZB* hzb, float extentScale, const ChordInstanceCullingView& iv,
                        int bObjectUseLastFrameProject, ChordCountAndCmd in, ChordCountAndCmd* out)
{
    if (!hzb || !hzb->minTexels || !in.count || !in.cmds || !out) return fail(c, CHORDVIS_E_INVALID, "hzb_culling_generic: invalid HZB or command list");
    ChordCtx* k = c->depthCtx;
    if (in.cmds == k->lists[1].cmds) return fail(c, CHORDVIS_E_INVALID, "hzb_culling_generic: input aliases the output list");
    double mainCam[3], viewCam[3];
    double3_of(c->hView.iv.cameraWorldPos, mainCam);      // perView.cameraWorldPos (renderer.cpp:251-263 copies it into the main view's info)
    double3_of(iv.cameraWorldPos, viewCam);
    const float rel[3] = {(float)(mainCam[0] - viewCam[0]), (float)(mainCam[1] - viewCam[1]), (float)(mainCam[2] - viewCam[2])};
    HzbBuffers hb;
    hb.desc = hzb->desc; hb.minTexels = hzb->minTexels; hb.maxTexels = hzb->maxTexels; hb.validRange = hzb->validRange; hb.valid = true;
    CmdList inL; inL.count = in.count; inL.cmds = in.cmds; inL.capacity = in.capacity;
    k->hView.flags = c->hView.flags;
    CHORD_HIP(c, hipMemsetAsync(k->lists[1].count, 0, sizeof(uint32_t), k->stream));
    launch_hzb_cull_generic(k, hb, iv, rel, extentScale, bObjectUseLastFrameProject != 0, inL, k->lists[1]);
    CHORD_HIP(c, hipGetLastError());
    *out = k->lists[1].handle();
    return CHORDVIS_OK;
}

int chordvis_render_mesh_depth(ChordCtx* c, uint32_t instanceViewOffset, int bDepthClamped, float depthBiasConst, float depthBiasSlope,
                               ChordCountAndCmd in, ChordDepthTarget* out)
{
    int rc = need_child(c, "render_mesh_depth");
    if (rc) return rc;
    ChordCtx* k = c->depthCtx;
    if (instanceViewOffset >= c->dDepthImages.size()) return fail(c, CHORDVIS_E_INVALID, "render_mesh_depth: view beyond allocate_depth_views");
    if (k->depthViewCurrent != (int)instanceViewOffset)
        return fail(c, CHORDVIS_E_INVALID, "render_mesh_depth: chordvis_instance_culling_view(instanceViewOffset) must come first (it sets up the view's object matrices)");
    // queue.clearDepthStencil(depth, 0.0) + the depth-only draw (mesh_raster.cpp:500-501, 159-206).  Like the main view's first
    // pass the raster clears by writing every tile (no 33 MB memset of the words), and the tile kernel's fused tile-out writes
    // the D32 image itself and reduces the tile to HZB mips 0..5 of the child's chain 0 (round 2: memset, global-atomic
    // tile-out, an extract pass over the words, and for buildHZB an expand pass + the two mip kernels: four more passes over
    // the image per cascade).
    if (!k->inFrame || k->rasterCalls != 0) {            // (not the first depth pass since chordvis_instance_culling_view zeroed the state)
        CHORD_HIP(c, hipMemsetAsync(k->dCounters, 0, sizeof(DeviceCounters), k->stream));
        k->rasterCalls = 0; k->inFrame = false;
    }
    c->fusedDepthView = -1;
    if (in.count && in.cmds) {
        k->pendingClear = true;
        k->fuseHzb = true; k->fuseHzbSlot = 0; k->fuseHzbTemp = false;
        k->depthOutTarget = c->dDepthImages[instanceViewOffset];
        k->depthOnly = true; k->depthClamp = bDepthClamped != 0; k->depthBiasConst = depthBiasConst; k->depthBiasSlope = depthBiasSlope;
        rc = chordvis_render_mesh(k, in);
        k->depthOnly = false; k->depthClamp = false; k->depthBiasConst = 0.0f; k->depthBiasSlope = 0.0f;
        k->fuseHzb = false; k->depthOutTarget = nullptr; k->pendingClear = false; k->inFrame = false;
        if (rc) return child_fail(c, rc);
        c->fusedDepthView = (int)instanceViewOffset;
    } else {
        CHORD_HIP(c, hipMemsetAsync(c->dDepthImages[instanceViewOffset], 0, sizeof(float) * (size_t)k->width * k->height, k->stream));   // {nullptr, nullptr}: nothing to draw
    }
    if (out) { out->depth = c->dDepthImages[instanceViewOffset]; out->width = k->width; out->height = k->height; }
    return CHORDVIS_OK;
}

int chordvis_build_hzb_from_depth(ChordCtx* c, const ChordDepthTarget* depth, ChordHZB* out)
{
    int rc = need_child(c, "build_hzb_from_depth");
    if (rc) return rc;
    ChordCtx* k = c->depthCtx;
    if (!depth || !depth->depth || depth->width != k->width || depth->height != k->height)
        return fail(c, CHORDVIS_E_INVALID, "build_hzb_from_depth: not a depth target of this context's depth views");
    k->viewSet = true;
    if (c->fusedDepthView >= 0 && (size_t)c->fusedDepthView < c->dDepthImages.size() && depth->depth == c->dDepthImages[c->fusedDepthView]) {
        // the image came straight out of chordvis_render_mesh_depth: levels 0..5 of its min chain are already in chain 0
        // (tile kernel); the one-block tail finishes it -- the values chordvis_build_hzb computes from the image
        launch_hzb_tail(k, k->hzb[0], false, false);
        CHORD_HIP(c, hipGetLastError());
        if (out) { *out = k->hzb[0].handle(); out->maxTexels = nullptr; out->validRange = nullptr; }
        return CHORDVIS_OK;
    }
    // any other image of these views (a cached cascade): the HZB kernels read the depth half of 64-bit words, so the image
    // goes (back) into the child's words first
    c->fusedDepthView = -1;
    launch_depth_expand(k, depth->depth, (unsigned long long*)k->dVis, (size_t)k->width * k->height);
    CHORD_HIP(c, hipGetLastError());
    return child_fail(c, chordvis_build_hzb(k, 1, 0, 0, 0, out));
}

int chordvis_readback_depth(ChordCtx* c, const ChordDepthTarget* depth, float* host)
{
    if (!c || !depth || !depth->depth || !host) return fail(c, CHORDVIS_E_INVALID, "readback_depth: null argument");
    CHORD_HIP(c, hipStreamSynchronize(c->stream));
    CHORD_HIP(c, hipMemcpy(host, depth->depth, sizeof(float) * (size_t)depth->width * depth->height, hipMemcpyDeviceToHost));
    return CHORDVIS_OK;
}

// renderShadow -- mesh_raster.cpp:331-546.  The cascades' depth images and views live in the context from call to call
// (CascadeShadowHistory, extractCascadeShadowHistory :548-563); a cascade whose cache is valid this tick is left alone.
int chordvis_render_shadow(ChordCtx* c, const ChordCascadeConfig* cfg, const float lightDir[3], const uint32_t validDepthMinMax[2],
                           uint32_t tickCount, int bHzbCulling, ChordDepthTarget* outDepths, ChordInstanceCullingView* outViews,
                           uint32_t* outRenderedMask)
{
    if (!c || !cfg || !lightDir) return fail(c, CHORDVIS_E_INVALID, "render_shadow: null argument");
    if (!c->viewSet || !c->sceneLoaded) return fail(c, CHORDVIS_E_INVALID, "render_shadow: upload_scene and set_view (the main camera) must come first");
    if (cfg->cascadeCount < 1 || (uint32_t)cfg->cascadeCount > CHORD_MAX_CASCADES || cfg->realtimeCascadeCount < 0 || cfg->realtimeCascadeCount > cfg->cascadeCount)
        return fail(c, CHORDVIS_E_INVALID, "render_shadow: cascade counts out of range");
    const uint32_t count = (uint32_t)cfg->cascadeCount, realtime = (uint32_t)cfg->realtimeCascadeCount;
    int rc;
    // bCacheValid (:367-371): depth images exist, same direction, same config
    const bool sameCfg = c->shadowHistoryValid && std::memcmp(&c->shadowHistoryConfig, cfg, sizeof(*cfg)) == 0 &&
                         std::memcmp(c->shadowHistoryDir, lightDir, sizeof(float) * 3) == 0 &&
                         c->depthCtx && c->depthDim == cfg->cascadeDim && c->dDepthImages.size() == count;
    const bool bCacheValid = sameCfg;
    if (!c->depthCtx || c->depthDim != cfg->cascadeDim || c->dDepthImages.size() != count) {
        if ((rc = chordvis_allocate_depth_views(c, cfg->cascadeDim, count))) return rc;
    }
    auto cacheValid = [&](uint32_t cascadeId) {                                       // isCascadeCacheValid, cascade_setup.hlsl:8-22
        if (!bCacheValid || cascadeId < realtime) return false;
        return (tickCount % (count - realtime)) != (cascadeId - realtime);
    };
    // cascade setup pass (:417-441): last frame's views stay in place where the cache holds
    std::vector<ChordInstanceCullingView> historyViews = c->instanceViews;
    if (c->instanceViews.size() != count) c->instanceViews.assign(count, ChordInstanceCullingView{});
    if ((rc = chordvis_cascade_setup(cfg, &c->hView.view, &c->hView.iv, lightDir, validDepthMinMax, tickCount, bCacheValid ? 1 : 0, c->instanceViews.data())))
        return fail(c, rc, "render_shadow: cascade setup");
    ChordHZB prevHzb;
    std::memset(&prevHzb, 0, sizeof(prevHzb));
    bool havePrev = false;
    uint32_t prevCascade = 0, rendered = 0;
    for (int32_t cascadeId = (int32_t)count - 1; cascadeId >= 0; cascadeId--) {       // :443-531
        if (cacheValid((uint32_t)cascadeId)) continue;
        ChordCountAndCmd list;
        if ((rc = chordvis_instance_culling_view(c, (uint32_t)cascadeId, &list))) return rc;
        if (!havePrev) {
            // the first cascade rendered this tick: cull against the HZB of its own cached depth, seen from last frame's view (:457-480)
            if (bCacheValid && bHzbCulling && historyViews.size() == count) {
                ChordDepthTarget hist = {c->dDepthImages[cascadeId], c->depthDim, c->depthDim};
                ChordHZB hzb;
                if ((rc = chordvis_build_hzb_from_depth(c, &hist, &hzb))) return rc;
                if ((rc = generic_cull(c, &hzb, 1.5f, historyViews[cascadeId], 0, list, &list))) return rc;   // sShadowExtentScaleForHZBCulling, :35
            }
        } else if (bHzbCulling) {
            if ((rc = generic_cull(c, &prevHzb, 1.5f, c->instanceViews[prevCascade], 0, list, &list))) return rc;   // :484-497
        }
        ChordDepthTarget target;
        if ((rc = chordvis_render_mesh_depth(c, (uint32_t)cascadeId, 1, cfg->shadowBiasConst, cfg->shadowBiasSlope, list, &target))) return rc;
        rendered |= 1u << cascadeId;
        if (cascadeId != 0) {                                                         // :525-529
            if ((rc = chordvis_build_hzb_from_depth(c, &target, &prevHzb))) return rc;
            havePrev = true; prevCascade = (uint32_t)cascadeId;
        }
    }
    c->shadowHistoryValid = true;                                                     // extractCascadeShadowHistory
    c->shadowHistoryConfig = *cfg;
    std::memcpy(c->shadowHistoryDir, lightDir, sizeof(float) * 3);
    if (outDepths) for (uint32_t i = 0; i < count; i++) outDepths[i] = ChordDepthTarget{c->dDepthImages[i], c->depthDim, c->depthDim};
    if (outViews) std::memcpy(outViews, c->instanceViews.data(), sizeof(ChordInstanceCullingView) * count);
    if (outRenderedMask) *outRenderedMask = rendered;
    return CHORDVIS_OK;
}

int chordvis_depth_view_stats(ChordCtx* c, ChordStats* out)
{
    int rc = need_child(c, "depth_view_stats");
    if (rc) return rc;
    return child_fail(c, chordvis_stats(c->depthCtx, out));
}

} // extern "C"
