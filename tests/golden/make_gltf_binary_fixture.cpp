// Fixture generator for the reference's GLTFBinary asset container (test infrastructure; runs only in the build container).
// Compiles the reference's VENDORED cereal and LZ4 where they lie (/root/reference/external) and archives a small asset the
// way the reference does:
//   saveAsset<T>            source/asset/serialize.h:217-270   (cereal binary of T -> optional LZ4 -> cereal binary of meta + string)
//   AssetCompressedMeta     serialize.h:202-215                 (rawSize, compressionSize, compressionMode as size_t)
//   GLTFBinary              serialize.h:84-99, asset_gltf.h:260-300
//   GLTFMeshlet / GLTFMeshletGroup / GLTFBVHNode   serialize.h:47-75 (member by member), wrappers asset_gltf.h:174-191
//   glm adapters            source/pch.h:100-121                (component by component)
//   class versions          utils.h:119-123 registerPODClassMember -> CEREAL_CLASS_VERSION(T, kAssetVersion), kAssetVersion = 0
//                           (asset_common.cpp:6-13)
// The structs below mirror those declarations (same members, same archive order); nothing of the reference is copied into the
// repository -- the outputs are the bytes cereal and LZ4 produce and a JSON of the values that went in.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include <glm/glm.hpp>
#include <cereal/access.hpp>
#include <cereal/cereal.hpp>
#include <cereal/types/string.hpp>
#include <cereal/types/vector.hpp>
#include <cereal/archives/binary.hpp>
#include <lz4.h>

namespace glm {
template <class Archive> void serialize(Archive& archive, glm::vec2& v) { archive(v.x, v.y); }
template <class Archive> void serialize(Archive& archive, glm::vec3& v) { archive(v.x, v.y, v.z); }
template <class Archive> void serialize(Archive& archive, glm::vec4& v) { archive(v.x, v.y, v.z, v.w); }
}

namespace mirror {
struct GPUBVHNode { glm::vec4 sphere; uint32_t children[8]; uint32_t bvhNodeCount, leafMeshletGroupOffset, leafMeshletGroupCount; };
struct GPUGLTFMeshletGroup { glm::vec3 clusterPosCenter; float parentError; glm::vec3 parentPosCenter; float error; uint32_t meshletOffset, meshletCount; };
struct GPUGLTFMeshlet { glm::vec3 posMin; uint32_t dataOffset; glm::vec3 posMax; uint32_t vertexTriangleCount; glm::vec3 coneAxis; float coneCutOff; glm::vec3 coneApex; uint32_t lod; };

struct GLTFBVHNode {
    GPUBVHNode data;
    template <class Ar> void serialize(Ar& ar, std::uint32_t const)
    { ar(data.sphere); ar(data.children); ar(data.leafMeshletGroupOffset); ar(data.leafMeshletGroupCount); ar(data.bvhNodeCount); }
};
struct GLTFMeshletGroup {
    GPUGLTFMeshletGroup data;
    template <class Ar> void serialize(Ar& ar, std::uint32_t const)
    { ar(data.clusterPosCenter); ar(data.parentError); ar(data.parentPosCenter); ar(data.error); ar(data.meshletOffset); ar(data.meshletCount); }
};
struct GLTFMeshlet {
    GPUGLTFMeshlet data;
    template <class Ar> void serialize(Ar& ar, std::uint32_t const)
    { ar(data.posMin); ar(data.dataOffset); ar(data.posMax); ar(data.vertexTriangleCount); ar(data.coneCutOff); ar(data.coneAxis); ar(data.coneApex); ar(data.lod); }
};
struct GLTFBinary {
    struct PrimitiveDatas {
        std::vector<GLTFMeshlet> meshlets; std::vector<uint32_t> meshletDatas; std::vector<GLTFBVHNode> bvhNodes;
        std::vector<GLTFMeshletGroup> meshletGroups; std::vector<uint32_t> meshletGroupIndices; std::vector<uint32_t> lod0Indices;
        std::vector<glm::vec3> positions, normals; std::vector<glm::vec2> texcoords0; std::vector<glm::vec4> tangents;
        std::vector<glm::vec2> texcoords1; std::vector<glm::vec4> colors0; std::vector<glm::vec3> smoothNormals;
    } primitiveData;
    template <class Ar> void serialize(Ar& ar, std::uint32_t const)
    {
        ar(primitiveData.positions); ar(primitiveData.normals); ar(primitiveData.texcoords0); ar(primitiveData.tangents);
        ar(primitiveData.smoothNormals); ar(primitiveData.texcoords1); ar(primitiveData.colors0);
        ar(primitiveData.meshlets); ar(primitiveData.meshletDatas); ar(primitiveData.bvhNodes); ar(primitiveData.meshletGroups);
        ar(primitiveData.meshletGroupIndices); ar(primitiveData.lod0Indices);
    }
};
enum class ECompressionMode { None, Lz4, MAX };
struct AssetCompressedMeta {
    ECompressionMode compressionMode; int32_t rawSize; int32_t compressionSize;
    template <class Ar> void serialize(Ar& ar)
    { ar(rawSize, compressionSize); { size_t e = (size_t)compressionMode; ar(cereal::make_nvp("enum__type__compressionMode", e)); compressionMode = (ECompressionMode)e; } }
};
}
CEREAL_CLASS_VERSION(mirror::GLTFBVHNode, 0);
CEREAL_CLASS_VERSION(mirror::GLTFMeshletGroup, 0);
CEREAL_CLASS_VERSION(mirror::GLTFMeshlet, 0);
CEREAL_CLASS_VERSION(mirror::GLTFBinary, 0);

static uint32_t pcg(uint32_t v) { uint32_t s = v * 747796405u + 2891336453u; uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u; return (w >> 22u) ^ w; }
static float rnd(uint32_t i) { return (float)(pcg(i) & 0xFFFFFFu) / 16777216.0f * 4.0f - 2.0f; }

static bool save(const mirror::GLTFBinary& in, mirror::ECompressionMode mode, const char* path)
{
    std::string raw;
    { std::stringstream ss; cereal::BinaryOutputArchive ar(ss); ar(in); raw = ss.str(); }
    mirror::AssetCompressedMeta meta; meta.compressionMode = mode; meta.rawSize = (int32_t)raw.size();
    std::string comp;
    if (mode == mirror::ECompressionMode::Lz4) {
        comp.resize(LZ4_compressBound((int)raw.size()));
        meta.compressionSize = LZ4_compress_default(raw.c_str(), comp.data(), (int)raw.size(), (int)comp.size());
        comp.resize(meta.compressionSize);
    } else { meta.compressionSize = meta.rawSize; comp = raw; }
    std::ofstream os(path, std::ios::binary);
    cereal::BinaryOutputArchive ar(os);
    ar(meta, comp);
    return true;
}

// --check FILE: reads an archive (e.g. one the LIBRARY wrote) the way loadAsset does (serialize.h:272-320) -- real cereal, real
// LZ4_decompress_safe -- and prints what it holds; tests/test_nanite_builder.py runs it where the reference exists.
static int check(const char* path)
{
    mirror::AssetCompressedMeta meta; std::string comp;
    { std::ifstream is(path, std::ios::binary); cereal::BinaryInputArchive ar(is); ar(meta, comp); }
    if ((size_t)meta.compressionSize != comp.size()) { printf("BAD sizes\n"); return 1; }
    std::string raw;
    if (meta.compressionMode == mirror::ECompressionMode::Lz4) {
        raw.resize(meta.rawSize);
        const int n = LZ4_decompress_safe(comp.data(), raw.data(), meta.compressionSize, meta.rawSize);
        if (n != meta.rawSize) { printf("BAD lz4 %d\n", n); return 1; }
    } else raw = comp;
    mirror::GLTFBinary b;
    { std::stringstream ss; ss << raw; cereal::BinaryInputArchive ar(ss); ar(b); }
    const auto& d = b.primitiveData;
    double sum = 0; for (const auto& p : d.positions) sum += (double)p.x + 2.0 * p.y + 3.0 * p.z;
    unsigned long long h = 1469598103934665603ull;
    auto mix = [&](uint32_t v) { h = (h ^ v) * 1099511628211ull; };
    for (const auto& m : d.meshlets) { mix(m.data.dataOffset); mix(m.data.vertexTriangleCount); mix(m.data.lod); uint32_t u; memcpy(&u, &m.data.coneCutOff, 4); mix(u); memcpy(&u, &m.data.coneAxis.x, 4); mix(u); }
    for (uint32_t v : d.meshletDatas) mix(v);
    for (const auto& n : d.bvhNodes) { mix(n.data.bvhNodeCount); mix(n.data.leafMeshletGroupOffset); mix(n.data.leafMeshletGroupCount); mix(n.data.children[7]); }
    for (const auto& g : d.meshletGroups) { mix(g.data.meshletOffset); mix(g.data.meshletCount); uint32_t u; memcpy(&u, &g.data.parentError, 4); mix(u); }
    for (uint32_t v : d.meshletGroupIndices) mix(v);
    printf("OK mode %d raw %d positions %zu texcoords0 %zu meshlets %zu meshletDatas %zu bvhNodes %zu groups %zu groupIndices %zu possum %.9g hash %llu\n",
           (int)meta.compressionMode, meta.rawSize, d.positions.size(), d.texcoords0.size(), d.meshlets.size(), d.meshletDatas.size(), d.bvhNodes.size(),
           d.meshletGroups.size(), d.meshletGroupIndices.size(), sum, h);
    return 0;
}

int main(int argc, char** argv)
{
    if (argc > 2 && std::string(argv[1]) == "--check") return check(argv[2]);
    const int V = argc > 1 ? atoi(argv[1]) : 150, M = argc > 2 ? atoi(argv[2]) : 5;
    mirror::GLTFBinary b; auto& d = b.primitiveData; uint32_t k = 1;
    for (int i = 0; i < V; i++) {
        d.positions.push_back({rnd(k), rnd(k + 1), rnd(k + 2)}); d.normals.push_back({rnd(k + 3), rnd(k + 4), rnd(k + 5)});
        d.texcoords0.push_back({rnd(k + 6), rnd(k + 7)}); d.tangents.push_back({rnd(k + 8), rnd(k + 9), rnd(k + 10), 1.0f}); k += 11;
    }
    for (int m = 0; m < M; m++) {
        mirror::GLTFMeshlet x; x.data.posMin = {rnd(k), rnd(k + 1), rnd(k + 2)}; x.data.posMax = {rnd(k + 3), rnd(k + 4), rnd(k + 5)};
        x.data.dataOffset = (uint32_t)d.meshletDatas.size(); const uint32_t nv = 8 + m, nt = 6 + 2 * m; x.data.vertexTriangleCount = nv | (nt << 8);
        x.data.coneAxis = {rnd(k + 6), rnd(k + 7), rnd(k + 8)}; x.data.coneCutOff = rnd(k + 9); x.data.coneApex = {rnd(k + 10), rnd(k + 11), rnd(k + 12)}; x.data.lod = m & 1; k += 13;
        for (uint32_t i = 0; i < nv; i++) d.meshletDatas.push_back(pcg(k++) % V);
        for (uint32_t i = 0; i < nt; i++) { uint32_t a = pcg(k++) % nv, c = pcg(k++) % nv, e = pcg(k++) % nv; d.meshletDatas.push_back(a | (c << 8) | (e << 16)); }
        d.meshlets.push_back(x);
    }
    for (int g = 0; g < 2; g++) {
        mirror::GLTFMeshletGroup x; x.data.clusterPosCenter = {rnd(k), rnd(k + 1), rnd(k + 2)}; x.data.parentError = g ? 3.4028235e38f : 0.25f;
        x.data.parentPosCenter = {rnd(k + 3), rnd(k + 4), rnd(k + 5)}; x.data.error = g ? 0.125f : -1.0f; x.data.meshletOffset = g * 3; x.data.meshletCount = g ? M - 3 : 3; k += 6;
        d.meshletGroups.push_back(x);
    }
    for (int i = 0; i < M; i++) d.meshletGroupIndices.push_back((uint32_t)i);
    { mirror::GLTFBVHNode n; n.data.sphere = {rnd(k), rnd(k + 1), rnd(k + 2), 2.5f}; for (int i = 0; i < 8; i++) n.data.children[i] = ~0u; n.data.bvhNodeCount = 1; n.data.leafMeshletGroupOffset = 0; n.data.leafMeshletGroupCount = 2; d.bvhNodes.push_back(n); }
    for (int i = 0; i < 12; i++) d.lod0Indices.push_back(pcg(k++) % V);
    save(b, mirror::ECompressionMode::None, argc > 3 ? argv[3] : "gltf_binary_raw.bin");
    save(b, mirror::ECompressionMode::Lz4, argc > 4 ? argv[4] : "gltf_binary_lz4.bin");
    // the values, for the test
    FILE* f = fopen(argc > 5 ? argv[5] : "gltf_binary.json", "w");
    fprintf(f, "{\"generator\": \"tests/golden/make_gltf_binary_fixture.sh (reference's vendored cereal + lz4)\", \"vertexCount\": %d,\n \"positions\": [", V);
    for (size_t i = 0; i < d.positions.size(); i++) fprintf(f, "%s%.9g, %.9g, %.9g", i ? ", " : "", d.positions[i].x, d.positions[i].y, d.positions[i].z);
    fprintf(f, "],\n \"texcoords0\": [");
    for (size_t i = 0; i < d.texcoords0.size(); i++) fprintf(f, "%s%.9g, %.9g", i ? ", " : "", d.texcoords0[i].x, d.texcoords0[i].y);
    fprintf(f, "],\n \"meshletDatas\": [");
    for (size_t i = 0; i < d.meshletDatas.size(); i++) fprintf(f, "%s%u", i ? ", " : "", d.meshletDatas[i]);
    fprintf(f, "],\n \"meshletGroupIndices\": [");
    for (size_t i = 0; i < d.meshletGroupIndices.size(); i++) fprintf(f, "%s%u", i ? ", " : "", d.meshletGroupIndices[i]);
    fprintf(f, "],\n \"meshlets\": [");
    for (size_t i = 0; i < d.meshlets.size(); i++) { const auto& m = d.meshlets[i].data;
        fprintf(f, "%s[%.9g, %.9g, %.9g, %u, %.9g, %.9g, %.9g, %u, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %u]", i ? ", " : "", m.posMin.x, m.posMin.y, m.posMin.z, m.dataOffset,
                m.posMax.x, m.posMax.y, m.posMax.z, m.vertexTriangleCount, m.coneAxis.x, m.coneAxis.y, m.coneAxis.z, m.coneCutOff, m.coneApex.x, m.coneApex.y, m.coneApex.z, m.lod); }
    fprintf(f, "],\n \"meshletGroups\": [");
    for (size_t i = 0; i < d.meshletGroups.size(); i++) { const auto& g = d.meshletGroups[i].data;
        fprintf(f, "%s[%.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %u, %u]", i ? ", " : "", g.clusterPosCenter.x, g.clusterPosCenter.y, g.clusterPosCenter.z, g.parentError,
                g.parentPosCenter.x, g.parentPosCenter.y, g.parentPosCenter.z, g.error, g.meshletOffset, g.meshletCount); }
    fprintf(f, "],\n \"bvhNodes\": [");
    for (size_t i = 0; i < d.bvhNodes.size(); i++) { const auto& n = d.bvhNodes[i].data;
        fprintf(f, "%s[%.9g, %.9g, %.9g, %.9g, %u, %u, %u, %u, %u, %u, %u, %u, %u, %u, %u]", i ? ", " : "", n.sphere.x, n.sphere.y, n.sphere.z, n.sphere.w, n.children[0], n.children[1], n.children[2], n.children[3],
                n.children[4], n.children[5], n.children[6], n.children[7], n.bvhNodeCount, n.leafMeshletGroupOffset, n.leafMeshletGroupCount); }
    fprintf(f, "]}\n");
    fclose(f);
    return 0;
}
