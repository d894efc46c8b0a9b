#!/bin/sh
# Regenerates meshopt_bounds.json with the reference's vendored meshoptimizer (only present in the build container): the
# reference's own sources are compiled where they lie, nothing of them is copied.  Meshes: the bumpy sphere of the builder
# tests (scenes.bumpy_sphere_mesh(96, 1): 18 k triangles -> ~150 meshlets), a finer one (~100 more), and a unit cube (one
# meshlet whose normals span every direction: the degenerate cone).
set -e
cd "$(dirname "$0")"
REF=/root/reference/source/asset/meshoptimizer
g++ -O2 -std=c++17 -I $REF make_meshopt_fixture.cpp $REF/meshopt_clusterizer.cpp $REF/meshopt_allocator.cpp -o /tmp/make_meshopt_fixture
dump() { python3 - "$@" <<'PY'
import sys, numpy as np
sys.path.insert(0, "../..")
from chord_amd import scenes
kind = sys.argv[1]
if kind == "cube":
    p = np.array([[x, y, z] for x in (0, 1) for y in (0, 1) for z in (0, 1)], np.float32)
    q = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    i = np.array([[a, b, c, a, c, d] for a, b, c, d in q], np.uint32).reshape(-1)
else:
    p, i, _ = scenes.bumpy_sphere_mesh(int(sys.argv[2]), int(sys.argv[3]))
print(len(p), len(i))
print(" ".join("%.9g" % v for v in p.reshape(-1)))
print(" ".join(str(v) for v in i))
PY
}
{
  echo '{"generator": "tests/golden/make_meshopt_fixture.sh (reference meshoptimizer 0.21: meshopt_buildMeshlets + meshopt_computeMeshletBounds)", "meshes": ['
  dump sphere 96 1 | /tmp/make_meshopt_fixture bumpy_sphere_96 160
  echo ','
  dump sphere 64 2 | /tmp/make_meshopt_fixture bumpy_sphere_64 80
  echo ','
  dump cube | /tmp/make_meshopt_fixture cube 4
  echo ']}'
} > meshopt_bounds.json
python3 -c "import json; d = json.load(open('meshopt_bounds.json')); print('wrote meshopt_bounds.json:', [(m['name'], len(m['meshlets'])) for m in d['meshes']])"
