// Fixture generator: meshlet bounds and normal cones as the REFERENCE's vendored meshoptimizer 0.21 computes them
// (source/asset/meshoptimizer/meshopt_clusterizer.cpp: meshopt_buildMeshlets + meshopt_computeMeshletBounds, called exactly
// as source/asset/nanite_builder.cpp:443-486 calls them: 255 vertices / 128 triangles per meshlet, cone weight 0.7,
// meshopt_optimizeMeshlet in between).  Built and run in the build container only (make_meshopt_fixture.sh compiles the
// reference's sources where they lie); what is committed is its OUTPUT, tests/golden/meshopt_bounds.json: per meshlet the
// geometry (vertex positions, local triangle indices) and the reference's bounds for it.
//
// stdin: "<vertexCount> <indexCount>\n" then the positions (3 floats per vertex) and the indices as text; a mesh per run.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "meshoptimizer.h"

int main(int argc, char** argv)
{
    const char* name = argc > 1 ? argv[1] : "mesh";
    const size_t limit = argc > 2 ? (size_t)atoi(argv[2]) : 1000000;
    unsigned vc = 0, ic = 0;
    if (scanf("%u %u", &vc, &ic) != 2) return 1;
    std::vector<float> pos((size_t)vc * 3);
    std::vector<unsigned> idx(ic);
    for (float& f : pos) if (scanf("%f", &f) != 1) return 1;
    for (unsigned& i : idx) if (scanf("%u", &i) != 1) return 1;
    const size_t maxV = 255, maxT = 128;                                  // kNaniteMeshletMaxVertices / MaxTriangle, base.h:429-430
    std::vector<meshopt_Meshlet> meshlets(meshopt_buildMeshletsBound(idx.size(), maxV, maxT));
    std::vector<unsigned> mv(meshlets.size() * maxV);
    std::vector<unsigned char> mt(meshlets.size() * maxT * 3);
    meshlets.resize(meshopt_buildMeshlets(meshlets.data(), mv.data(), mt.data(), idx.data(), idx.size(), pos.data(), vc, 12, maxV, maxT, 0.7f));
    printf("{\"name\": \"%s\", \"meshlets\": [\n", name);
    size_t emitted = 0;
    for (size_t m = 0; m < meshlets.size() && emitted < limit; m++) {
        const meshopt_Meshlet& ml = meshlets[m];
        meshopt_optimizeMeshlet(&mv[ml.vertex_offset], &mt[ml.triangle_offset], ml.triangle_count, ml.vertex_count);
        const meshopt_Bounds b = meshopt_computeMeshletBounds(&mv[ml.vertex_offset], &mt[ml.triangle_offset], ml.triangle_count, pos.data(), vc, 12);
        printf("%s{\"positions\": [", emitted ? ",\n" : "");
        for (unsigned v = 0; v < ml.vertex_count; v++) {
            const unsigned g = mv[ml.vertex_offset + v];
            printf("%s%.9g, %.9g, %.9g", v ? ", " : "", pos[3 * g], pos[3 * g + 1], pos[3 * g + 2]);
        }
        printf("], \"triangles\": [");
        for (unsigned t = 0; t < ml.triangle_count * 3; t++) printf("%s%u", t ? ", " : "", (unsigned)mt[ml.triangle_offset + t]);
        printf("], \"center\": [%.9g, %.9g, %.9g], \"radius\": %.9g, \"cone_apex\": [%.9g, %.9g, %.9g], \"cone_axis\": [%.9g, %.9g, %.9g], \"cone_cutoff\": %.9g}",
               b.center[0], b.center[1], b.center[2], b.radius, b.cone_apex[0], b.cone_apex[1], b.cone_apex[2],
               b.cone_axis[0], b.cone_axis[1], b.cone_axis[2], b.cone_cutoff);
        emitted++;
    }
    printf("\n]}\n");
    return 0;
}
