#!/bin/sh
# Regenerates gltf_binary_raw.bin / gltf_binary_lz4.bin / gltf_binary.json with the reference's vendored cereal and LZ4 (only
# present in the build container): their sources are compiled where they lie, nothing of them is copied.
set -e
cd "$(dirname "$0")"
REF=/root/reference/external
g++ -O2 -std=c++17 -I $REF/include -I $REF/lz4/lib make_gltf_binary_fixture.cpp $REF/lz4/lib/lz4.c -o /tmp/make_gltf_binary_fixture
/tmp/make_gltf_binary_fixture 150 5 gltf_binary_raw.bin gltf_binary_lz4.bin gltf_binary.json
ls -l gltf_binary_raw.bin gltf_binary_lz4.bin gltf_binary.json
