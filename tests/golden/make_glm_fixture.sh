#!/bin/sh
# Regenerates glm_camera.json with the reference's vendored glm (only present in the build container).
set -e
cd "$(dirname "$0")"
g++ -O0 -std=c++17 -ffp-contract=off -I /root/reference/external/include make_glm_fixture.cpp -o /tmp/make_glm_fixture
/tmp/make_glm_fixture > glm_camera.json
echo "wrote $(pwd)/glm_camera.json"
