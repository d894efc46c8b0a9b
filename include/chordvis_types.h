/*
 * chordvis_types.h — the data contract of the visibility hot path.
 *
 * Plain-C restatement of the POD records the reference shares between its C++
 * host code and its HLSL kernels.  Byte layouts are identical to the reference
 * so a maintainer can hand the reference's own arrays across the C ABI:
 *
 *   ChordMeshlet              <- GPUGLTFMeshlet            install/resource/shader/gltf.h:38-51
 *   ChordMeshletGroup         <- GPUGLTFMeshletGroup       install/resource/shader/gltf.h:26-36
 *   ChordPrimitive            <- GLTFPrimitiveBuffer       install/resource/shader/gltf.h:65-91
 *   ChordMaterial             <- GLTFMaterialGPUData       install/resource/shader/gltf.h:118-153
 *   ChordObjectBasicData      <- GPUObjectBasicData        install/resource/shader/base.h:343-351
 *   ChordObject               <- GPUObjectGLTFPrimitive    install/resource/shader/base.h:353-360
 *   ChordInstanceCullingView  <- InstanceCullingViewInfo   install/resource/shader/base.h:121-135
 *   ChordDrawCmd              <- uint3 draw command        install/resource/shader/instance_culling.hlsl:28-33
 *
 * Matrices are glm column-major in memory (element (row r, col c) at m[c*4+r]);
 * the kernels read them with the HLSL convention M[r][c] == glm m[c][r]
 * (install/resource/shader/hzb_culling_generic.hlsl:78-80).
 *
 * What changes versus the reference: bindless buffer ids
 * (GLTFPrimitiveDatasBuffer, gltf.h:94-116) become plain pointers in
 * ChordAssetDesc, and the slice of PerframeCameraView (base.h:292-340) that the
 * path reads is restated as ChordCameraView.
 */
#ifndef CHORDVIS_TYPES_H
#define CHORDVIS_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* switchFlags bits — gltf.h:10-12 */
#define CHORD_FLAG_FRUSTUM_CULL   (1u << 0) /* kFrustumCullingEnableBit  */
#define CHORD_FLAG_HZB_CULL       (1u << 1) /* kHZBCullingEnableBit      */
#define CHORD_FLAG_CONE_CULL      (1u << 2) /* kMeshletConeCullEnableBit */

/* base.h:428-436 */
#define CHORD_MESHLET_MAX_VERTICES     255u
#define CHORD_MESHLET_MAX_TRIANGLES    128u
#define CHORD_GROUP_MAX_MESHLETS       4u
#define CHORD_HZB_MAX_MIPS             12u
#define CHORD_MAX_INSTANCE_ID          0xFFFFFFu /* kMaxInstanceIdCount, base.h:412 */

/* nanite_shared.hlsli:11-12 */
#define CHORD_ERROR_PIXEL_THRESHOLD    1.0f
#define CHORD_ERROR_RADIUS_ROOT        3e38f

/* base.h:442-444: sub-pixel precision of the raster the reference assumes */
#define CHORD_SUBPIXEL_BITS            8

typedef struct ChordMat4 { float m[16]; } ChordMat4;

typedef struct ChordMeshlet {
    float    posMin[3];
    uint32_t dataOffset;           /* u32 index into meshletData           */
    float    posMax[3];
    uint32_t vertexTriangleCount;  /* V & 0xff | T << 8   (gltf.h:60-62)   */
    float    coneAxis[3];
    float    coneCutOff;
    float    coneApex[3];
    uint32_t lod;
} ChordMeshlet;

typedef struct ChordMeshletGroup {
    float    clusterPosCenter[3];
    float    parentError;          /* FLT_MAX when un-parented (root)      */
    float    parentPosCenter[3];
    float    error;                /* -1 at LOD0                           */
    uint32_t meshletOffset;        /* into meshletGroupIndices             */
    uint32_t meshletCount;         /* <= 4                                 */
} ChordMeshletGroup;

/* GPUBVHNode -- gltf.h:16-24, built by buildBVHTree / flattenBVH (nanite_builder.cpp:215-416): an 8-wide tree over the
 * PARENTED cluster groups of a primitive, keyed on their parent-error spheres; `sphere` bounds the parent spheres of
 * every group in the node's subtree (nanite_builder.cpp:53-56), the root's own leaves are the un-parented groups.
 * Indices are relative to the primitive's first node / first group; nodes are in breadth-first order and the
 * primitive's groups in the order the nodes list them.  The reference uploads the tree and never walks it
 * (instance_culling.hlsl:96-99); chordvis_set_cull_mode(ctx, 1) does. */
#define CHORD_BVH_WIDTH 8u             /* kNaniteBVHLevelNodeCount, base.h:433 */
#define CHORD_BVH_MAX_LEVELS 14u       /* kNaniteMaxBVHLevelCount,  base.h:432 */
#define CHORD_BVH_NO_CHILD 0xFFFFFFFFu
typedef struct ChordBVHNode {
    float    sphere[4];
    uint32_t children[8];
    uint32_t bvhNodeCount;             /* nodes of the subtree, this one included */
    uint32_t leafMeshletGroupOffset;
    uint32_t leafMeshletGroupCount;
} ChordBVHNode;

typedef struct ChordPrimitive {
    float    posMin[3];
    uint32_t primitiveDatasBufferId;   /* index into ChordSceneDesc.assets */
    float    posMax[3];
    uint32_t vertexOffset;
    float    posAverage[3];
    uint32_t vertexCount;
    uint32_t meshletOffset;
    uint32_t color0Offset;
    uint32_t smoothNormalOffset;
    uint32_t textureCoord1Offset;
    uint32_t bvhNodeOffset;
    uint32_t meshletGroupOffset;
    uint32_t meshletGroupIndicesOffset;
    uint32_t meshletGroupCount;
    uint32_t lod0IndicesOffset;
    uint32_t lod0IndicesCount;
    uint32_t pad0;
    uint32_t pad1;
} ChordPrimitive;

typedef struct ChordMaterial {
    uint32_t alphaMode;
    float    alphaCutOff;
    uint32_t bTwoSided;
    uint32_t baseColorId;
    float    baseColorFactor[4];
    float    emissiveFactor[3];
    uint32_t emissiveTexture;
    float    metallicFactor;
    float    roughnessFactor;
    uint32_t metallicRoughnessTexture;
    uint32_t normalTexture;
    uint32_t baseColorSampler;
    uint32_t emissiveSampler;
    uint32_t normalSampler;
    uint32_t metallicRoughnessSampler;
    float    normalFactorScale;
    uint32_t bExistOcclusion;
    float    occlusionTextureStrength;
    uint32_t materialType;
} ChordMaterial;

typedef struct ChordObjectBasicData {
    ChordMat4 localToTranslatedWorld;
    ChordMat4 translatedWorldToLocal;
    ChordMat4 localToTranslatedWorldLastFrame;
    float     scaleExtractFromMatrix[4]; /* .w = max |scale| */
} ChordObjectBasicData;

typedef struct ChordObject {
    ChordObjectBasicData basicData;
    uint32_t GLTFPrimitiveDetail;  /* index into primitives */
    uint32_t GLTFMaterialData;     /* index into materials  */
    uint32_t pad1;
    uint32_t pad2;
} ChordObject;

typedef struct ChordInstanceCullingView {
    ChordMat4 translatedWorldToClip;
    ChordMat4 clipToTranslatedWorld;
    uint32_t  cameraWorldPos[8];          /* GPUStorageDouble4 */
    float     orthoDepthConvertToView[4];
    float     renderDimension[4];         /* w, h, 1/w, 1/h */
    float     frustumPlanesRS[6][4];      /* left, down, right, top, front, back (camera.h:10-18) */
} ChordInstanceCullingView;

/* The slice of PerframeCameraView (base.h:292-340) the path reads, plus the
 * one host-precomputed scalar the canonical arithmetic needs (lodScale). */
typedef struct ChordCameraView {
    ChordMat4 translatedWorldToView;
    ChordMat4 translatedWorldToClip;
    ChordMat4 translatedWorldToClipLastFrame;
    float     renderDimension[4];         /* w, h, 1/w, 1/h */
    float     cameraFovy;
    float     zNear;
    float     zFar;
    float     lodScale;                   /* (h * 0.5f) / tanf(0.5f * fovy), base.hlsli:503-518 */
    ChordMat4 clipToTranslatedWorldWithZFar_NoJitter;   /* camera.cpp:25-31: inverse of (perspectiveRH_ZO(fovy, aspect, zFar, zNear) * view);
                                                         * read by the cascade setup only (cascade_setup.hlsl:175) */
} ChordCameraView;

/* CascadeShadowMapConfig -- render_helper.h:462-484 (the fields the cascade setup and the depth passes read) */
typedef struct ChordCascadeConfig {
    int32_t  cascadeCount;            /* 8 */
    int32_t  realtimeCascadeCount;    /* 3 */
    uint32_t cascadeDim;              /* 2048 */
    float    cascadeStartDistance;    /* 0 */
    float    cascadeEndDistance;      /* 80 */
    float    farCascadeEndDistance;   /* 800 */
    float    splitLambda;             /* 0.8 */
    float    farCascadeSplitLambda;   /* 0.8 */
    float    shadowBiasConst;         /* 0 */
    float    shadowBiasSlope;         /* 0 */
    float    radiusScaleFixed;        /* 10 */
} ChordCascadeConfig;
#define CHORD_MAX_CASCADES 32u        /* the setup shader's group size (cascade_setup.hlsl:79) */

typedef struct ChordDrawCmd {
    uint32_t objectId;
    uint32_t meshletId;    /* index into the asset's meshlet array (primitive.meshletOffset applied) */
    uint32_t slot;         /* index in the post-instanceCulling list; the visibility payload */
} ChordDrawCmd;

/* GLTFMaterialGPUData::alphaMode -- EAlphaMode (asset_gltf.h:161; asset_gltf_material.cpp:703-705).  The visibility and
 * depth passes draw the opaque and the masked bucket; blended materials are in neither (mesh_raster.cpp:178,224). */
#define CHORD_ALPHA_OPAQUE 0u
#define CHORD_ALPHA_MASK   1u
#define CHORD_ALPHA_BLEND  2u

/* GLTFSampler (asset_gltf.h:9-39), the glTF enum values as the reference keeps them. */
#define CHORD_FILTER_NEAREST                 9728u
#define CHORD_FILTER_LINEAR                  9729u
#define CHORD_FILTER_NEAREST_MIPMAP_NEAREST  9984u
#define CHORD_FILTER_LINEAR_MIPMAP_NEAREST   9985u
#define CHORD_FILTER_NEAREST_MIPMAP_LINEAR   9986u
#define CHORD_FILTER_LINEAR_MIPMAP_LINEAR    9987u
#define CHORD_WRAP_REPEAT           10497u
#define CHORD_WRAP_CLAMP_TO_EDGE    33071u
#define CHORD_WRAP_MIRRORED_REPEAT  33648u
typedef struct ChordSampler {
    uint32_t minFilter, magFilter, wrapS, wrapT;
} ChordSampler;

/* A base-colour texture as the masked buckets read it (mesh_raster.hlsl:198-204: only .w of the sample is used):
 * RGBA8, `mipCount` levels back to back starting with level 0 (level l is max(1, width >> l) x max(1, height >> l)).
 * ChordMaterial::baseColorId / baseColorSampler index ChordSceneDesc::textures / samplers (the reference's bindless ids);
 * an id >= textureCount reads as the reference's white fallback (alpha 1, asset_gltf.cpp:374). */
typedef struct ChordTexture {
    const uint8_t* rgba8;
    uint32_t width, height, mipCount, pad;
} ChordTexture;

/* One GLTFPrimitiveDatasBuffer (gltf.h:94-116): bindless ids -> host pointers. */
typedef struct ChordAssetDesc {
    const ChordMeshlet*      meshlets;           uint32_t meshletCount;
    const ChordMeshletGroup* meshletGroups;      uint32_t meshletGroupCount;
    const uint32_t*          meshletGroupIndices;uint32_t meshletGroupIndexCount;
    const uint32_t*          meshletData;        uint32_t meshletDataCount;  /* u32 words */
    const float*             positions;          uint32_t vertexCount;       /* float3 tightly packed */
    const float*             texcoord0;          uint32_t texcoord0Count;    /* float2 per vertex (textureCoord0Buffer), or NULL / 0:
                                                                              * masked materials then sample at uv (0, 0) */
    const ChordBVHNode*      bvhNodes;           uint32_t bvhNodeCount;      /* bvhNodeBuffer, or NULL / 0 (flat culling only);
                                                                              * primitive p's tree starts at bvhNodeOffset */
} ChordAssetDesc;

typedef struct ChordSceneDesc {
    const ChordObject*    objects;    uint32_t objectCount;
    const ChordPrimitive* primitives; uint32_t primitiveCount;
    const ChordMaterial*  materials;  uint32_t materialCount;
    const ChordAssetDesc* assets;     uint32_t assetCount;
    const ChordTexture*   textures;   uint32_t textureCount;   /* may be NULL / 0 */
    const ChordSampler*   samplers;   uint32_t samplerCount;   /* may be NULL / 0: REPEAT, NEAREST */
} ChordSceneDesc;

/* Visibility texel — base.hlsli:437-447 (low 32 bits) under the depth bits
 * (high 32 bits, D32 reverse-Z of render_textures.h:25). 0 == empty. */
static inline uint32_t chord_encode_triangle_instance(uint32_t triangleId, uint32_t instanceId)
{
    return (((instanceId + 1u) & CHORD_MAX_INSTANCE_ID) << 8) | (triangleId & 0xFFu);
}
static inline void chord_decode_triangle_instance(uint32_t pack, uint32_t* triangleId, uint32_t* instanceId)
{
    *triangleId = pack & 0xFFu;
    *instanceId = ((pack >> 8) & CHORD_MAX_INSTANCE_ID) - 1u;
}

/* HZB chain geometry — hzb.cpp:49-63.  mip l has (w0 >> l, h0 >> l) texels
 * (min 1), stored as IEEE binary16, mips back to back, each row-major. */
typedef struct ChordHZBDesc {
    uint32_t srcWidth, srcHeight;
    uint32_t width, height;        /* mip 0 extent */
    uint32_t mipCount;
    uint32_t mipOffset[CHORD_HZB_MAX_MIPS]; /* in texels from the chain base */
    uint32_t totalTexels;
} ChordHZBDesc;

#ifdef __cplusplus
}
static_assert(sizeof(ChordMeshlet) == 64, "GPUGLTFMeshlet");
static_assert(sizeof(ChordMeshletGroup) == 40, "GPUGLTFMeshletGroup");
static_assert(sizeof(ChordBVHNode) == 60, "GPUBVHNode");
static_assert(sizeof(ChordPrimitive) == 96, "GLTFPrimitiveBuffer");
static_assert(sizeof(ChordMaterial) == 96, "GLTFMaterialGPUData");
static_assert(sizeof(ChordObjectBasicData) == 208, "GPUObjectBasicData");
static_assert(sizeof(ChordObject) == 224, "GPUObjectGLTFPrimitive");
static_assert(sizeof(ChordInstanceCullingView) == 288, "InstanceCullingViewInfo");
static_assert(sizeof(ChordDrawCmd) == 12, "uint3 draw cmd");
#endif

#endif /* CHORDVIS_TYPES_H */
