"""numpy mirrors of include/chordvis_types.h (the reference's shared C++/HLSL records).

Byte layouts follow install/resource/shader/gltf.h:16-153 and base.h:121-135,343-360
of the reference; sizes are asserted below and again against the C header in tests.
"""
import ctypes as C

import numpy as np

f32, u32 = np.float32, np.uint32

MESHLET = np.dtype([
    ("posMin", f32, 3), ("dataOffset", u32),
    ("posMax", f32, 3), ("vertexTriangleCount", u32),
    ("coneAxis", f32, 3), ("coneCutOff", f32),
    ("coneApex", f32, 3), ("lod", u32),
])
MESHLET_GROUP = np.dtype([
    ("clusterPosCenter", f32, 3), ("parentError", f32),
    ("parentPosCenter", f32, 3), ("error", f32),
    ("meshletOffset", u32), ("meshletCount", u32),
])
BVH_NODE = np.dtype([
    ("sphere", f32, 4), ("children", u32, 8), ("bvhNodeCount", u32), ("leafMeshletGroupOffset", u32), ("leafMeshletGroupCount", u32),
])
PRIMITIVE = np.dtype([
    ("posMin", f32, 3), ("primitiveDatasBufferId", u32),
    ("posMax", f32, 3), ("vertexOffset", u32),
    ("posAverage", f32, 3), ("vertexCount", u32),
    ("meshletOffset", u32), ("color0Offset", u32), ("smoothNormalOffset", u32), ("textureCoord1Offset", u32),
    ("bvhNodeOffset", u32), ("meshletGroupOffset", u32), ("meshletGroupIndicesOffset", u32), ("meshletGroupCount", u32),
    ("lod0IndicesOffset", u32), ("lod0IndicesCount", u32), ("pad0", u32), ("pad1", u32),
])
MATERIAL = np.dtype([
    ("alphaMode", u32), ("alphaCutOff", f32), ("bTwoSided", u32), ("baseColorId", u32),
    ("baseColorFactor", f32, 4),
    ("emissiveFactor", f32, 3), ("emissiveTexture", u32),
    ("metallicFactor", f32), ("roughnessFactor", f32), ("metallicRoughnessTexture", u32), ("normalTexture", u32),
    ("baseColorSampler", u32), ("emissiveSampler", u32), ("normalSampler", u32), ("metallicRoughnessSampler", u32),
    ("normalFactorScale", f32), ("bExistOcclusion", u32), ("occlusionTextureStrength", f32), ("materialType", u32),
])
OBJECT = np.dtype([
    ("localToTranslatedWorld", f32, 16),
    ("translatedWorldToLocal", f32, 16),
    ("localToTranslatedWorldLastFrame", f32, 16),
    ("scaleExtractFromMatrix", f32, 4),
    ("GLTFPrimitiveDetail", u32), ("GLTFMaterialData", u32), ("pad1", u32), ("pad2", u32),
])
INSTANCE_CULLING_VIEW = np.dtype([
    ("translatedWorldToClip", f32, 16),
    ("clipToTranslatedWorld", f32, 16),
    ("cameraWorldPos", u32, 8),
    ("orthoDepthConvertToView", f32, 4),
    ("renderDimension", f32, 4),
    ("frustumPlanesRS", f32, (6, 4)),
])
CAMERA_VIEW = np.dtype([
    ("translatedWorldToView", f32, 16),
    ("translatedWorldToClip", f32, 16),
    ("translatedWorldToClipLastFrame", f32, 16),
    ("renderDimension", f32, 4),
    ("cameraFovy", f32), ("zNear", f32), ("zFar", f32), ("lodScale", f32),
    ("clipToTranslatedWorldWithZFar_NoJitter", f32, 16),
])
CASCADE_CONFIG = np.dtype([
    ("cascadeCount", np.int32), ("realtimeCascadeCount", np.int32), ("cascadeDim", u32),
    ("cascadeStartDistance", f32), ("cascadeEndDistance", f32), ("farCascadeEndDistance", f32), ("splitLambda", f32),
    ("farCascadeSplitLambda", f32), ("shadowBiasConst", f32), ("shadowBiasSlope", f32), ("radiusScaleFixed", f32),
])


def default_cascade_config(**kw):
    """CascadeShadowMapConfig defaults (render_helper.h:467-483)."""
    c = np.zeros(1, dtype=CASCADE_CONFIG)
    c["cascadeCount"], c["realtimeCascadeCount"], c["cascadeDim"] = 8, 3, 2048
    c["cascadeStartDistance"], c["cascadeEndDistance"], c["farCascadeEndDistance"] = 0.0, 80.0, 800.0
    c["splitLambda"], c["farCascadeSplitLambda"], c["radiusScaleFixed"] = 0.8, 0.8, 10.0
    for k, v in kw.items():
        c[k] = v
    return c
DRAW_CMD = np.dtype([("objectId", u32), ("meshletId", u32), ("slot", u32)])

assert MESHLET.itemsize == 64 and MESHLET_GROUP.itemsize == 40 and PRIMITIVE.itemsize == 96 and BVH_NODE.itemsize == 60
assert MATERIAL.itemsize == 96 and OBJECT.itemsize == 224 and INSTANCE_CULLING_VIEW.itemsize == 288
assert CAMERA_VIEW.itemsize == 288 and DRAW_CMD.itemsize == 12 and CASCADE_CONFIG.itemsize == 44

FLAG_FRUSTUM_CULL = 1 << 0
FLAG_HZB_CULL = 1 << 1
FLAG_CONE_CULL = 1 << 2

HZB_MAX_MIPS = 12


class AssetDesc(C.Structure):
    _fields_ = [
        ("meshlets", C.c_void_p), ("meshletCount", C.c_uint32),
        ("meshletGroups", C.c_void_p), ("meshletGroupCount", C.c_uint32),
        ("meshletGroupIndices", C.c_void_p), ("meshletGroupIndexCount", C.c_uint32),
        ("meshletData", C.c_void_p), ("meshletDataCount", C.c_uint32),
        ("positions", C.c_void_p), ("vertexCount", C.c_uint32),
        ("texcoord0", C.c_void_p), ("texcoord0Count", C.c_uint32),
        ("bvhNodes", C.c_void_p), ("bvhNodeCount", C.c_uint32),
    ]


class SceneDesc(C.Structure):
    _fields_ = [
        ("objects", C.c_void_p), ("objectCount", C.c_uint32),
        ("primitives", C.c_void_p), ("primitiveCount", C.c_uint32),
        ("materials", C.c_void_p), ("materialCount", C.c_uint32),
        ("assets", C.POINTER(AssetDesc)), ("assetCount", C.c_uint32),
        ("textures", C.c_void_p), ("textureCount", C.c_uint32),
        ("samplers", C.c_void_p), ("samplerCount", C.c_uint32),
    ]


class Texture(C.Structure):
    _fields_ = [("rgba8", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32), ("mipCount", C.c_uint32), ("pad", C.c_uint32)]


SAMPLER = np.dtype([("minFilter", u32), ("magFilter", u32), ("wrapS", u32), ("wrapT", u32)])
ALPHA_OPAQUE, ALPHA_MASK, ALPHA_BLEND = 0, 1, 2
FILTER_NEAREST, FILTER_LINEAR = 9728, 9729
FILTER_NEAREST_MIPMAP_NEAREST, FILTER_LINEAR_MIPMAP_NEAREST, FILTER_NEAREST_MIPMAP_LINEAR, FILTER_LINEAR_MIPMAP_LINEAR = 9984, 9985, 9986, 9987
WRAP_REPEAT, WRAP_CLAMP_TO_EDGE, WRAP_MIRRORED_REPEAT = 10497, 33071, 33648


def mip_chain_rgba8(level0):
    """All mip levels of an (H, W, 4) uint8 image back to back (2x2 box filter, floor division; odd sizes drop the last
    row / column like a plain downsample), as ChordTexture::rgba8 wants them.  Returns (bytes array, mipCount)."""
    img = np.ascontiguousarray(level0, dtype=np.uint8)
    levels = [img]
    while img.shape[0] > 1 or img.shape[1] > 1:
        h, w = max(1, img.shape[0] // 2), max(1, img.shape[1] // 2)
        a = img[:h * 2 if img.shape[0] > 1 else 1, :w * 2 if img.shape[1] > 1 else 1].astype(np.uint32)
        if img.shape[0] > 1:
            a = a[0::2] + a[1::2]
        else:
            a = a * 2
        if img.shape[1] > 1:
            a = a[:, 0::2] + a[:, 1::2]
        else:
            a = a * 2
        img = ((a + 2) // 4).astype(np.uint8)
        levels.append(img)
    return np.concatenate([l.reshape(-1) for l in levels]), len(levels)


class HZBDesc(C.Structure):
    _fields_ = [
        ("srcWidth", C.c_uint32), ("srcHeight", C.c_uint32),
        ("width", C.c_uint32), ("height", C.c_uint32),
        ("mipCount", C.c_uint32),
        ("mipOffset", C.c_uint32 * HZB_MAX_MIPS),
        ("totalTexels", C.c_uint32),
    ]

    def mip_dims(self, level):
        return max(1, self.width >> level), max(1, self.height >> level)

    def valid_dims(self, level):
        w, h = self.mip_dims(level)
        return (min(w, (((self.srcWidth - 1) >> 1) >> level) + 1),
                min(h, (((self.srcHeight - 1) >> 1) >> level) + 1))


class Scene:
    """Host-side scene: the reference's per-frame collected arrays, one asset.

    Holds numpy arrays alive and exposes a ctypes SceneDesc over them.
    """

    def __init__(self, objects, primitives, materials, meshlets, groups, group_indices, meshlet_data, positions,
                 name="scene", texcoord0=None, textures=(), samplers=None, bvh_nodes=None):
        """textures: sequence of (H, W, 4) uint8 images (mip chains are built here); samplers: SAMPLER records."""
        self.name = name
        self.objects = np.ascontiguousarray(objects, dtype=OBJECT)
        self.primitives = np.ascontiguousarray(primitives, dtype=PRIMITIVE)
        self.materials = np.ascontiguousarray(materials, dtype=MATERIAL)
        self.meshlets = np.ascontiguousarray(meshlets, dtype=MESHLET)
        self.groups = np.ascontiguousarray(groups, dtype=MESHLET_GROUP)
        self.group_indices = np.ascontiguousarray(group_indices, dtype=u32)
        self.meshlet_data = np.ascontiguousarray(meshlet_data, dtype=u32)
        self.positions = np.ascontiguousarray(positions, dtype=f32).reshape(-1, 3)
        self.bvh_nodes = None if bvh_nodes is None else np.ascontiguousarray(bvh_nodes, dtype=BVH_NODE)
        self.texcoord0 = None if texcoord0 is None else np.ascontiguousarray(texcoord0, dtype=f32).reshape(-1, 2)
        self.texture_images = list(textures)
        self._tex_chains = [mip_chain_rgba8(t) for t in self.texture_images]
        self._textures = (Texture * max(1, len(self._tex_chains)))()
        for i, (chain, mips) in enumerate(self._tex_chains):
            self._textures[i] = Texture(chain.ctypes.data, self.texture_images[i].shape[1], self.texture_images[i].shape[0], mips, 0)
        self.samplers = np.zeros(0, dtype=SAMPLER) if samplers is None else np.ascontiguousarray(samplers, dtype=SAMPLER)
        self._asset = AssetDesc(
            self.meshlets.ctypes.data, len(self.meshlets),
            self.groups.ctypes.data, len(self.groups),
            self.group_indices.ctypes.data, len(self.group_indices),
            self.meshlet_data.ctypes.data, len(self.meshlet_data),
            self.positions.ctypes.data, len(self.positions),
            self.texcoord0.ctypes.data if self.texcoord0 is not None else None, len(self.texcoord0) if self.texcoord0 is not None else 0,
            self.bvh_nodes.ctypes.data if self.bvh_nodes is not None else None, len(self.bvh_nodes) if self.bvh_nodes is not None else 0,
        )
        self._assets = (AssetDesc * 1)(self._asset)
        self.desc = SceneDesc(
            self.objects.ctypes.data, len(self.objects),
            self.primitives.ctypes.data, len(self.primitives),
            self.materials.ctypes.data, len(self.materials),
            self._assets, 1,
            C.cast(self._textures, C.c_void_p) if self._tex_chains else None, len(self._tex_chains),
            self.samplers.ctypes.data if len(self.samplers) else None, len(self.samplers),
        )

    def with_objects(self, objects=None, materials=None):
        """The same geometry, textures and samplers under other object / material records (shallow: arrays are shared)."""
        out = Scene(self.objects if objects is None else objects, self.primitives, self.materials if materials is None else materials,
                    self.meshlets, self.groups, self.group_indices, self.meshlet_data, self.positions, name=self.name,
                    texcoord0=self.texcoord0, textures=self.texture_images, samplers=self.samplers, bvh_nodes=self.bvh_nodes)
        if hasattr(self, "local_to_world"):
            out.local_to_world = self.local_to_world
        return out

    # --- aggregate counts the reference keeps in PerframeCollected (scene_common.h) -------------
    @property
    def object_count(self):
        return len(self.objects)

    @property
    def lod0_meshlet_instances(self):
        """Upper bound of draw commands: every meshlet instance (all LODs) of every object."""
        prim = self.primitives[self.objects["GLTFPrimitiveDetail"]]
        per_prim = np.zeros(len(self.primitives), dtype=np.int64)
        for i, p in enumerate(self.primitives):
            g = self.groups[p["meshletGroupOffset"]: p["meshletGroupOffset"] + p["meshletGroupCount"]]
            per_prim[i] = int(g["meshletCount"].sum())
        del prim
        return int(per_prim[self.objects["GLTFPrimitiveDetail"]].sum())

    @property
    def group_instances(self):
        return int(self.primitives["meshletGroupCount"][self.objects["GLTFPrimitiveDetail"]].sum())

    def triangle_count_lod0(self):
        lod0 = self.meshlets["lod"] == 0
        tri = (self.meshlets["vertexTriangleCount"] >> 8) & 0xFF
        per_meshlet = np.where(lod0, tri, 0).astype(np.int64)
        # meshlets are laid out per primitive [meshletOffset, next)
        cs = np.concatenate([[0], np.cumsum(per_meshlet)])
        offs = self.primitives["meshletOffset"].astype(np.int64)
        ends = np.concatenate([offs[1:], [len(self.meshlets)]])
        per_prim = cs[ends] - cs[offs]
        return int(per_prim[self.objects["GLTFPrimitiveDetail"]].sum())
