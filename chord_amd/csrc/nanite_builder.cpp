// Producer of the path's input format: triangle mesh -> meshlets, cluster groups with LOD errors, BVH.
//
// Mirrors NaniteBuilder (source/asset/nanite_builder.cpp): build() :882-921 = buildMeshlets (LOD 0, :432-536) -> up to 11 x
// MeshletsGMSS (:764-880: group the meshlets of a level, merge each group's triangles, simplify to half with the group's
// border locked, split into the next level's meshlets, hand the simplification error up as the children's parentError)
// -> buildBVHTree (:313-416); packing as asset_gltf_helper.cpp:496-548.  Host code, offline, no device involved.
//
// The reference delegates three steps to third-party code:
//   meshopt_buildMeshlets (vendored meshoptimizer 0.21, MIT)  -> clusterize(): own greedy growth over triangle adjacency
//   meshopt_computeMeshletBounds (same library, meshopt_clusterizer.cpp:792-845) -> meshlet_bounds(): the cone part FOLLOWS
//       that function (credit: meshoptimizer, (c) Arseny Kapoulkine, MIT licence) -- the `mindp <= 0.1` cut, cutoff =
//       sqrt(1 - mindp^2) and apex = center - axis * max(dc / dn) are its formulas and its names, and like it the axis is the
//       centre of a bounding sphere of the normals and the reference point the centre of one of the positions (own
//       implementation of Ritter's sphere, bounding_sphere()).  tests/test_nanite_builder.py checks the result against fixtures
//       produced by the vendored function itself (tests/golden/meshopt_bounds.json): safe against ground truth, never culls a
//       sampled camera the reference's cone keeps, axis within a stated angle of it
//   METIS_PartGraphKway (binary-only in the reference tree, version not recorded)          -> partition_groups(): greedy
//       graph growing by shared-edge weight into parts of min(n / 2, 4) meshlets
//   meshopt_simplifyWithAttributes (LockBorder | Sparse | ErrorAbsolute)                   -> simplify(): half-edge collapses
//       ordered by quadric error, border vertices of the group locked, no triangle flips, positions only
// So the OUTPUT differs from the reference's for the same mesh (parity unpinned for the builder); what is kept
// is the contract the runtime relies on: <= 255 vertices / 128 triangles per meshlet, <= 4 meshlets per group, LOD 0
// error -1, un-parented groups FLT_MAX, a parent's error >= every child's, BVH spheres around the parent-error spheres.

#include "../../include/chordvis.h"

#include <algorithm>
#include <array>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <queue>
#include <unordered_map>
#include <exception>
#include <vector>

namespace {

constexpr uint32_t kMaxVerts = CHORD_MESHLET_MAX_VERTICES, kMaxTris = CHORD_MESHLET_MAX_TRIANGLES;
constexpr uint32_t kMinNumMeshletPerGroup = 2, kMaxNumMeshletPerGroup = 4;                   // nanite_builder.cpp:15-16
constexpr float kGroupSimplifyThreshold = 0.5f, kGroupSimplifyMinReduce = 0.8f;             // :17-21
constexpr float kSimplifyErrorMin = 0.01f, kSimplifyErrorMax = 0.10f;                       // :24-25
constexpr uint32_t kMaxLODCount = 12;                                                       // kNaniteMaxLODCount, base.h:431

struct V3 { float x, y, z; };
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float len(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 vmin(V3 a, V3 b) { return {std::min(a.x, b.x), std::min(a.y, b.y), std::min(a.z, b.z)}; }
inline V3 vmax(V3 a, V3 b) { return {std::max(a.x, b.x), std::max(a.y, b.y), std::max(a.z, b.z)}; }

struct BMeshlet {
    std::vector<uint32_t> verts;       // global vertex ids (<= 255)
    std::vector<uint8_t> tris;         // local ids, 3 per triangle (<= 128 triangles)
    V3 posMin, posMax, coneAxis, coneApex; float coneCutOff;
    uint32_t lod; float error, parentError; V3 clusterPosCenter, parentPosCenter;
};

inline uint64_t edge_key(uint32_t a, uint32_t b) { return a < b ? ((uint64_t)b << 32) | a : ((uint64_t)a << 32) | b; }

// ---- buildMeshlets: greedy growth over the triangle adjacency ---------------------------------------------------------
// Ritter's approximate bounding sphere (1990): the most separated pair of the six axis-extremal points seeds the sphere,
// every point still outside grows it just enough.  Used for the meshlet's positions (cone apex reference point) and for its
// unit normals seen as points (the centre of THAT sphere is the cone axis: a minimal-cone estimate instead of the mean
// normal, which a few outlying triangles tilt away from the bulk).
struct Sphere { V3 c; float r; };
Sphere bounding_sphere(const std::vector<V3>& pts)
{
    Sphere s{{0, 0, 0}, 0.0f};
    if (pts.empty()) return s;
    size_t lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    for (size_t i = 0; i < pts.size(); i++) {
        const float v[3] = {pts[i].x, pts[i].y, pts[i].z};
        for (int k = 0; k < 3; k++) {
            const float l = k == 0 ? pts[lo[k]].x : k == 1 ? pts[lo[k]].y : pts[lo[k]].z;
            const float h = k == 0 ? pts[hi[k]].x : k == 1 ? pts[hi[k]].y : pts[hi[k]].z;
            if (v[k] < l) lo[k] = i;
            if (v[k] > h) hi[k] = i;
        }
    }
    int best = 0; float bestD = -1.0f;
    for (int k = 0; k < 3; k++) { const V3 d = pts[hi[k]] - pts[lo[k]]; const float d2 = dot(d, d); if (d2 > bestD) { bestD = d2; best = k; } }
    s.c = (pts[lo[best]] + pts[hi[best]]) * 0.5f;
    s.r = std::sqrt(bestD) * 0.5f;
    for (const V3& p : pts) {
        const V3 d = p - s.c;
        const float d2 = dot(d, d);
        if (d2 > s.r * s.r) {
            const float dl = std::sqrt(d2), k = 0.5f + (s.r / dl) * 0.5f;      // new centre between the far side of the old sphere and p
            s.c = s.c * k + p * (1.0f - k);
            s.r = (s.r + dl) * 0.5f;
        }
    }
    return s;
}

// Bounds + normal cone of a meshlet.  The cone follows meshopt_computeMeshletBounds (vendored meshoptimizer 0.21,
// meshopt_clusterizer.cpp:792-845, MIT): axis = centre of the normals' bounding sphere, `mindp <= 0.1` -> no cone, cutoff =
// sqrt(1 - mindp^2), apex = centre - axis * max(dc / dn) -- its formulas and names, see the header comment.
void meshlet_bounds(BMeshlet& m, const std::vector<V3>& pos)
{
    m.posMin = {FLT_MAX, FLT_MAX, FLT_MAX}; m.posMax = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (uint32_t v : m.verts) { m.posMin = vmin(m.posMin, pos[v]); m.posMax = vmax(m.posMax, pos[v]); }
    const size_t T = m.tris.size() / 3;
    std::vector<V3> normals, corners, used;
    for (size_t t = 0; t < T; t++) {
        const V3 a = pos[m.verts[m.tris[3 * t]]], b = pos[m.verts[m.tris[3 * t + 1]]], c = pos[m.verts[m.tris[3 * t + 2]]];
        V3 n = cross(b - a, c - a);
        const float l = len(n);
        if (l == 0.0f) continue;
        n = n * (1.0f / l);
        normals.push_back(n); corners.push_back(a);
        used.push_back(a); used.push_back(b); used.push_back(c);
    }
    m.coneAxis = {0, 0, 0}; m.coneApex = {0, 0, 0}; m.coneCutOff = 1.0f;          // degenerate: never culled (dot <= 1 is >= 1 only at equality)
    if (normals.empty()) return;
    const V3 center = bounding_sphere(used).c;
    V3 axis = bounding_sphere(normals).c;
    const float al = len(axis);
    if (al == 0.0f) return;
    axis = axis * (1.0f / al);
    float mindp = 1.0f;
    for (const V3& n : normals) mindp = std::min(mindp, dot(n, axis));
    if (mindp <= 0.1f) { m.coneAxis = axis; return; }                              // normals spread over more than ~84 degrees
    float maxt = 0.0f;                                                             // apex = center - axis * maxt puts every triangle's plane in front
    for (size_t i = 0; i < normals.size(); i++) {
        const float dc = dot(center - corners[i], normals[i]), dn = dot(axis, normals[i]);
        maxt = std::max(maxt, dc / dn);
    }
    m.coneAxis = axis;
    m.coneApex = center - axis * maxt;
    // (a hair wider than the exact bound: the runtime evaluates the test in fp32 after a matrix transform of the camera)
    m.coneCutOff = std::min(1.0f, std::sqrt(1.0f - mindp * mindp) + 1.0e-4f);
}

std::vector<BMeshlet> clusterize(const std::vector<V3>& pos, const std::vector<uint32_t>& indices, uint32_t lod, float error, V3 clusterCenter)
{
    const uint32_t T = (uint32_t)(indices.size() / 3);
    std::unordered_map<uint64_t, std::vector<uint32_t>> edgeTris;
    edgeTris.reserve(T * 2);
    for (uint32_t t = 0; t < T; t++)
        for (int e = 0; e < 3; e++) edgeTris[edge_key(indices[3 * t + e], indices[3 * t + (e + 1) % 3])].push_back(t);
    std::vector<uint8_t> used(T, 0);
    std::vector<BMeshlet> out;
    std::vector<int32_t> localOf(pos.size(), -1);                                  // vertex -> local id in the meshlet being grown
    uint32_t next = 0;
    for (;;) {
        while (next < T && used[next]) next++;
        if (next >= T) break;
        BMeshlet m;
        m.lod = lod; m.error = error; m.parentError = FLT_MAX; m.clusterPosCenter = clusterCenter; m.parentPosCenter = clusterCenter;
        std::vector<uint32_t> frontier;
        auto add = [&](uint32_t t) {
            used[t] = 1;
            for (int k = 0; k < 3; k++) {
                const uint32_t v = indices[3 * t + k];
                if (localOf[v] < 0) { localOf[v] = (int32_t)m.verts.size(); m.verts.push_back(v); }
                m.tris.push_back((uint8_t)localOf[v]);
            }
            for (int e = 0; e < 3; e++)
                for (uint32_t n : edgeTris[edge_key(indices[3 * t + e], indices[3 * t + (e + 1) % 3])]) if (!used[n]) frontier.push_back(n);
        };
        add(next);
        while (m.tris.size() / 3 < kMaxTris) {
            // the frontier triangle that brings the fewest new vertices (most shared), lowest index on ties
            int best = -1; uint32_t bestNew = 4, bestTri = 0;
            for (size_t i = 0; i < frontier.size();) {
                const uint32_t t = frontier[i];
                if (used[t]) { frontier[i] = frontier.back(); frontier.pop_back(); continue; }
                uint32_t nn = 0;
                for (int k = 0; k < 3; k++) nn += localOf[indices[3 * t + k]] < 0 ? 1u : 0u;
                if (nn < bestNew || (nn == bestNew && t < bestTri)) { best = (int)i; bestNew = nn; bestTri = t; }
                i++;
            }
            if (best < 0 || m.verts.size() + bestNew > kMaxVerts) break;
            add(bestTri);
        }
        for (uint32_t v : m.verts) localOf[v] = -1;
        meshlet_bounds(m, pos);
        out.push_back(std::move(m));
    }
    return out;
}

// ---- buildClusterGroup: partition of the meshlet adjacency graph (the reference calls METIS) ------------------------
std::vector<std::vector<uint32_t>> partition_groups(const std::vector<BMeshlet>& ms)
{
    const uint32_t n = (uint32_t)ms.size();
    std::vector<std::vector<uint32_t>> groups;
    if (n < kMinNumMeshletPerGroup) return groups;                                 // :588-593 -> the caller stops
    const uint32_t groupSize = std::min(n / kMinNumMeshletPerGroup, kMaxNumMeshletPerGroup);   // :595
    std::unordered_map<uint64_t, std::vector<uint32_t>> edge2m;
    for (uint32_t i = 0; i < n; i++)
        for (size_t t = 0; t < ms[i].tris.size() / 3; t++)
            for (int e = 0; e < 3; e++) {
                auto& v = edge2m[edge_key(ms[i].verts[ms[i].tris[3 * t + e]], ms[i].verts[ms[i].tris[3 * t + (e + 1) % 3]])];
                if (v.empty() || v.back() != i) v.push_back(i);
            }
    std::vector<std::map<uint32_t, uint32_t>> adj(n);                              // neighbour -> shared edges (the METIS edge weight, :660-676)
    bool anyShared = false;
    for (auto& kv : edge2m) {
        auto& v = kv.second;
        std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end());
        for (size_t a = 0; a < v.size(); a++) for (size_t b = a + 1; b < v.size(); b++) { adj[v[a]][v[b]]++; adj[v[b]][v[a]]++; anyShared = true; }
    }
    if (!anyShared) return groups;                                                 // :634-639
    std::vector<int32_t> part(n, -1);
    for (;;) {
        // seed: the unassigned meshlet with the fewest unassigned neighbours (peels the graph from its rim)
        int32_t seed = -1; uint32_t seedDeg = ~0u;
        for (uint32_t i = 0; i < n; i++) {
            if (part[i] >= 0) continue;
            uint32_t d = 0;
            for (auto& kv : adj[i]) d += part[kv.first] < 0 ? 1u : 0u;
            if (d < seedDeg) { seedDeg = d; seed = (int32_t)i; }
        }
        if (seed < 0) break;
        const int32_t g = (int32_t)groups.size();
        groups.push_back({(uint32_t)seed});
        part[seed] = g;
        while (groups[g].size() < groupSize) {
            std::map<uint32_t, uint32_t> w;                                        // unassigned neighbour -> connection to the group
            for (uint32_t mId : groups[g]) for (auto& kv : adj[mId]) if (part[kv.first] < 0) w[kv.first] += kv.second;
            if (w.empty()) break;
            uint32_t best = w.begin()->first;
            for (auto& kv : w) if (kv.second > w[best]) best = kv.first;
            groups[g].push_back(best);
            part[best] = g;
        }
    }
    // a part that could not grow (an island of the graph) joins the smallest part it touches, if that keeps it within 4
    for (size_t g = 0; g < groups.size(); g++) {
        if (groups[g].size() != 1) continue;
        const uint32_t mId = groups[g][0];
        int32_t target = -1;
        for (auto& kv : adj[mId]) {
            const int32_t og = part[kv.first];
            if (og >= 0 && og != (int32_t)g && groups[og].size() < kMaxNumMeshletPerGroup && (target < 0 || groups[og].size() < groups[target].size())) target = og;
        }
        if (target >= 0) { groups[target].push_back(mId); part[mId] = target; groups[g].clear(); }
    }
    groups.erase(std::remove_if(groups.begin(), groups.end(), [](const std::vector<uint32_t>& v) { return v.empty(); }), groups.end());
    return groups;
}

// ---- simplify: half-edge collapses by quadric error, border locked ---------------------------------------------------
struct Quadric { double a[10]; };   // symmetric 4x4: xx xy xz xw yy yz yw zz zw ww
inline void q_add_plane(Quadric& q, double nx, double ny, double nz, double d, double w)
{
    const double p[4] = {nx, ny, nz, d};
    int k = 0;
    for (int i = 0; i < 4; i++) for (int j = i; j < 4; j++) q.a[k++] += w * p[i] * p[j];
}
inline void q_add(Quadric& q, const Quadric& o) { for (int i = 0; i < 10; i++) q.a[i] += o.a[i]; }
inline double q_eval(const Quadric& q, V3 p)
{
    const double x = p.x, y = p.y, z = p.z;
    return q.a[0] * x * x + 2 * q.a[1] * x * y + 2 * q.a[2] * x * z + 2 * q.a[3] * x + q.a[4] * y * y + 2 * q.a[5] * y * z + 2 * q.a[6] * y +
           q.a[7] * z * z + 2 * q.a[8] * z + q.a[9];
}

// indices in / out (triangle list over global vertex ids); returns the largest collapse error as a distance
std::vector<uint32_t> simplify(const std::vector<V3>& pos, const std::vector<uint32_t>& in, size_t targetIndexCount, float targetError, float* outError)
{
    std::vector<uint32_t> idx = in;
    // compact vertex set of this group
    std::unordered_map<uint32_t, uint32_t> toLocal;
    std::vector<uint32_t> toGlobal;
    for (uint32_t v : idx) if (!toLocal.count(v)) { toLocal[v] = (uint32_t)toGlobal.size(); toGlobal.push_back(v); }
    const uint32_t V = (uint32_t)toGlobal.size();
    std::vector<std::array<uint32_t, 3>> tris;
    for (size_t t = 0; t < idx.size() / 3; t++) {
        std::array<uint32_t, 3> tr = {toLocal[idx[3 * t]], toLocal[idx[3 * t + 1]], toLocal[idx[3 * t + 2]]};
        if (tr[0] != tr[1] && tr[1] != tr[2] && tr[0] != tr[2]) tris.push_back(tr);
    }
    auto P = [&](uint32_t l) { return pos[toGlobal[l]]; };
    // border = vertices of edges used by exactly one triangle of the group (meshopt_SimplifyLockBorder)
    std::unordered_map<uint64_t, uint32_t> edgeUse;
    for (auto& tr : tris) for (int e = 0; e < 3; e++) edgeUse[edge_key(tr[e], tr[(e + 1) % 3])]++;
    std::vector<uint8_t> locked(V, 0);
    for (auto& kv : edgeUse) if (kv.second == 1) { locked[(uint32_t)kv.first] = 1; locked[(uint32_t)(kv.first >> 32)] = 1; }
    std::vector<Quadric> Q(V);
    for (auto& q : Q) std::memset(&q, 0, sizeof(q));
    for (auto& tr : tris) {
        const V3 a = P(tr[0]), b = P(tr[1]), c = P(tr[2]);
        V3 n = cross(b - a, c - a);
        const double area = len(n);
        if (area == 0.0) continue;
        n = n * (float)(1.0 / area);
        const double d = -dot(n, a);
        for (int k = 0; k < 3; k++) q_add_plane(Q[tr[k]], n.x, n.y, n.z, d, area);
    }
    std::vector<std::vector<uint32_t>> vtris(V);
    std::vector<uint8_t> dead(tris.size(), 0);
    for (uint32_t t = 0; t < tris.size(); t++) for (int k = 0; k < 3; k++) vtris[tris[t][k]].push_back(t);
    std::vector<uint32_t> remap(V);
    for (uint32_t i = 0; i < V; i++) remap[i] = i;
    std::vector<double> wsum(V, 0.0);                                              // area weight of each quadric (error -> squared distance)
    for (uint32_t t = 0; t < tris.size(); t++) { const double a = len(cross(P(tris[t][1]) - P(tris[t][0]), P(tris[t][2]) - P(tris[t][0]))); for (int k = 0; k < 3; k++) wsum[tris[t][k]] += a; }
    struct Cand { double err; uint32_t from, to, stamp; bool operator<(const Cand& o) const { return err > o.err; } };
    std::priority_queue<Cand> heap;
    std::vector<uint32_t> stamp(V, 0);
    auto push_edges = [&](uint32_t v) {
        if (locked[v]) return;                                                     // a locked vertex never moves (it may be collapsed ONTO)
        for (uint32_t t : vtris[v]) {
            if (dead[t]) continue;
            for (int k = 0; k < 3; k++) {
                const uint32_t u = tris[t][k];
                if (u == v) continue;
                Quadric q = Q[v]; q_add(q, Q[u]);
                const double w = wsum[v] + wsum[u];
                const double e = w > 0 ? std::max(0.0, q_eval(q, P(u))) / w : 0.0;
                heap.push({e, v, u, stamp[v]});
            }
        }
    };
    for (uint32_t v = 0; v < V; v++) push_edges(v);
    size_t liveTris = tris.size();
    double maxErr2 = 0.0;
    const double limit2 = (double)targetError * targetError;
    while (liveTris * 3 > targetIndexCount && !heap.empty()) {
        const Cand c = heap.top(); heap.pop();
        if (c.stamp != stamp[c.from] || remap[c.from] != c.from || remap[c.to] != c.to) continue;
        if (c.err > limit2) break;
        // no flips: every surviving triangle of `from` keeps the side its normal points to
        bool flip = false;
        for (uint32_t t : vtris[c.from]) {
            if (dead[t]) continue;
            const auto& tr = tris[t];
            if (tr[0] == c.to || tr[1] == c.to || tr[2] == c.to) continue;        // collapses away
            V3 p[3], q[3];
            for (int k = 0; k < 3; k++) { p[k] = P(tr[k]); q[k] = tr[k] == c.from ? P(c.to) : P(tr[k]); }
            const V3 n0 = cross(p[1] - p[0], p[2] - p[0]), n1 = cross(q[1] - q[0], q[2] - q[0]);
            if (dot(n0, n1) <= 0.0f) { flip = true; break; }
        }
        if (flip) continue;
        remap[c.from] = c.to;
        for (uint32_t t : vtris[c.from]) {
            if (dead[t]) continue;
            auto& tr = tris[t];
            bool hasTo = tr[0] == c.to || tr[1] == c.to || tr[2] == c.to;
            if (hasTo) { dead[t] = 1; liveTris--; continue; }
            for (int k = 0; k < 3; k++) if (tr[k] == c.from) tr[k] = c.to;
            vtris[c.to].push_back(t);
        }
        q_add(Q[c.to], Q[c.from]); wsum[c.to] += wsum[c.from];
        maxErr2 = std::max(maxErr2, c.err);
        stamp[c.to]++;
        push_edges(c.to);
        for (uint32_t t : vtris[c.to]) if (!dead[t]) for (int k = 0; k < 3; k++) if (tris[t][k] != c.to) { stamp[tris[t][k]]++; push_edges(tris[t][k]); }
    }
    std::vector<uint32_t> out;
    for (uint32_t t = 0; t < tris.size(); t++) if (!dead[t]) for (int k = 0; k < 3; k++) out.push_back(toGlobal[tris[t][k]]);
    if (outError) *outError = (float)std::sqrt(maxErr2);
    return out;
}

// ---- buildBVHTree (nanite_builder.cpp:77-416) ------------------------------------------------------------------------
struct BGroup { V3 clusterPosCenter; float error; V3 parentPosCenter; float parentError; std::vector<uint32_t> meshlets; };
struct TNode { V3 mn, mx; std::vector<uint32_t> leaves; int32_t children[8]; uint32_t depth; std::vector<uint32_t> todo; };

void build_tree(const std::vector<BGroup>& groups, std::vector<ChordBVHNode>& nodesOut, std::vector<uint32_t>& orderOut)
{
    std::vector<TNode> nodes(1);
    auto bounds = [&](const std::vector<uint32_t>& ids, V3& mn, V3& mx) {
        if (ids.empty()) { mn = {0, 0, 0}; mx = {0, 0, 0}; return; }
        mn = {FLT_MAX, FLT_MAX, FLT_MAX}; mx = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (uint32_t g : ids) {
            const V3 c = groups[g].parentPosCenter; const float r = groups[g].parentError;
            mn = vmin(mn, {c.x - r, c.y - r, c.z - r}); mx = vmax(mx, {c.x + r, c.y + r, c.z + r});
        }
    };
    auto longest = [](V3 mn, V3 mx) {                                              // :114-123
        const V3 d = mx - mn; uint32_t a = 0;
        if (d.y >= d.x && d.y >= d.z) a = 1;
        if (d.z >= d.x && d.z >= d.y) a = 2;
        return a;
    };
    auto halves = [&](const std::vector<uint32_t>& ids, V3 mn, V3 mx, std::vector<uint32_t> out[2]) {
        const uint32_t ax = longest(mn, mx);
        std::vector<uint32_t> s = ids;
        std::stable_sort(s.begin(), s.end(), [&](uint32_t a, uint32_t b) {
            const float* pa = &groups[a].parentPosCenter.x; const float* pb = &groups[b].parentPosCenter.x;
            return pa[ax] < pb[ax];
        });
        const size_t n = s.size();
        for (int i = 0; i < 2; i++) out[i].assign(s.begin() + i * n / 2, s.begin() + (i + 1) * n / 2);
    };
    TNode& root = nodes[0];
    for (int k = 0; k < 8; k++) root.children[k] = -1;
    root.depth = 0;
    for (uint32_t g = 0; g < groups.size(); g++) (groups[g].parentError < CHORD_ERROR_RADIUS_ROOT ? root.todo : root.leaves).push_back(g);
    bounds(root.todo, root.mn, root.mx);
    for (size_t qi = 0; qi < nodes.size(); qi++) {                                 // (index loop: `nodes` grows; breadth first like the std::queue of :86)
        std::vector<uint32_t> ids = std::move(nodes[qi].todo);
        if (ids.empty()) continue;
        if (ids.size() < CHORD_BVH_WIDTH || nodes[qi].depth == CHORD_BVH_MAX_LEVELS - 1) {   // :102
            nodes[qi].leaves.insert(nodes[qi].leaves.end(), ids.begin(), ids.end());
            continue;
        }
        std::vector<uint32_t> h0[2];
        halves(ids, nodes[qi].mn, nodes[qi].mx, h0);
        for (int i = 0; i < 2; i++) {
            V3 mn0, mx0; bounds(h0[i], mn0, mx0);
            std::vector<uint32_t> h1[2];
            halves(h0[i], mn0, mx0, h1);
            for (int j = 0; j < 2; j++) {
                V3 mn1, mx1; bounds(h1[j], mn1, mx1);
                std::vector<uint32_t> h2[2];
                halves(h1[j], mn1, mx1, h2);
                for (int k = 0; k < 2; k++) {
                    TNode ch;
                    bounds(h2[k], ch.mn, ch.mx);
                    for (int c = 0; c < 8; c++) ch.children[c] = -1;
                    ch.depth = nodes[qi].depth + 1; ch.todo = h2[k];
                    nodes[qi].children[(i * 2 + j) * 2 + k] = (int32_t)nodes.size();
                    nodes.push_back(std::move(ch));
                }
            }
        }
    }
    nodesOut.assign(nodes.size(), ChordBVHNode{});
    orderOut.clear();
    for (size_t n = 0; n < nodes.size(); n++) {                                    // flattenBVH :215-311 (the build order above IS breadth first)
        ChordBVHNode& o = nodesOut[n];
        const V3 c = (nodes[n].mx + nodes[n].mn) * 0.5f;
        o.sphere[0] = c.x; o.sphere[1] = c.y; o.sphere[2] = c.z; o.sphere[3] = 0.5f * len(nodes[n].mx - nodes[n].mn);   // sphereBuild :53-56
        for (int k = 0; k < 8; k++) o.children[k] = nodes[n].children[k] < 0 ? CHORD_BVH_NO_CHILD : (uint32_t)nodes[n].children[k];
        o.leafMeshletGroupOffset = (uint32_t)orderOut.size(); o.leafMeshletGroupCount = (uint32_t)nodes[n].leaves.size();
        orderOut.insert(orderOut.end(), nodes[n].leaves.begin(), nodes[n].leaves.end());
    }
    for (size_t n = nodes.size(); n-- > 0;) {
        uint32_t cnt = 1;
        for (int k = 0; k < 8; k++) if (nodes[n].children[k] >= 0) cnt += nodesOut[nodes[n].children[k]].bvhNodeCount;
        nodesOut[n].bvhNodeCount = cnt;
    }
}

} // namespace

struct ChordBuiltAsset {
    std::vector<float> positions, texcoords;
    std::vector<ChordMeshlet> meshlets;
    std::vector<ChordMeshletGroup> groups;
    std::vector<uint32_t> groupIndices, meshletData;
    std::vector<ChordBVHNode> bvh;
    ChordPrimitive prim;
    uint32_t lodCount = 0;
};

extern "C" {

// The bounds and normal cone the builder gives a meshlet, for a meshlet handed in as such (tests: against the reference's
// meshopt_computeMeshletBounds, tests/golden/meshopt_bounds.json).  positions: the meshlet's own vertices; triangles: 3 local
// indices per triangle.
int chordvis_meshlet_bounds(const float* positions, uint32_t vertexCount, const uint8_t* triangles, uint32_t triangleCount, ChordMeshlet* out)
{
    if (!positions || !triangles || !out || vertexCount == 0 || vertexCount > 255 || triangleCount == 0 || triangleCount > 128) return CHORDVIS_E_INVALID;
    std::vector<V3> pos(vertexCount);
    for (uint32_t v = 0; v < vertexCount; v++) pos[v] = {positions[3 * v], positions[3 * v + 1], positions[3 * v + 2]};
    BMeshlet m;
    m.verts.resize(vertexCount);
    for (uint32_t v = 0; v < vertexCount; v++) m.verts[v] = v;
    for (uint32_t t = 0; t < triangleCount * 3; t++) { if (triangles[t] >= vertexCount) return CHORDVIS_E_INVALID; m.tris.push_back(triangles[t]); }
    meshlet_bounds(m, pos);
    std::memset(out, 0, sizeof(*out));
    out->posMin[0] = m.posMin.x; out->posMin[1] = m.posMin.y; out->posMin[2] = m.posMin.z;
    out->posMax[0] = m.posMax.x; out->posMax[1] = m.posMax.y; out->posMax[2] = m.posMax.z;
    out->coneAxis[0] = m.coneAxis.x; out->coneAxis[1] = m.coneAxis.y; out->coneAxis[2] = m.coneAxis.z;
    out->coneApex[0] = m.coneApex.x; out->coneApex[1] = m.coneApex.y; out->coneApex[2] = m.coneApex.z;
    out->coneCutOff = m.coneCutOff;
    out->vertexTriangleCount = vertexCount | (triangleCount << 8);
    return CHORDVIS_OK;
}

int chordvis_nanite_build(const float* positionsIn, uint32_t vertexCount, const uint32_t* indicesIn, uint32_t indexCount,
                          const float* texcoord0, ChordBuiltAsset** out)
{
    if (!out) return CHORDVIS_E_INVALID;
    *out = nullptr;
    if (!positionsIn || !indicesIn || vertexCount == 0 || indexCount < 3 || indexCount % 3 != 0) return CHORDVIS_E_INVALID;   // "Nanite only support triangle mesh!" :885
    for (uint32_t i = 0; i < indexCount; i++) if (indicesIn[i] >= vertexCount) return CHORDVIS_E_INVALID;
    std::vector<V3> pos(vertexCount);
    std::memcpy(pos.data(), positionsIn, sizeof(V3) * vertexCount);
    std::vector<uint32_t> indices(indicesIn, indicesIn + indexCount);
    // meshopt_simplifyScale: the extent of the mesh
    V3 mn = {FLT_MAX, FLT_MAX, FLT_MAX}, mx = {-FLT_MAX, -FLT_MAX, -FLT_MAX}, avg = {0, 0, 0};
    for (const V3& p : pos) { mn = vmin(mn, p); mx = vmax(mx, p); avg = avg + p * (1.0f / (float)vertexCount); }
    const float scale = std::max(mx.x - mn.x, std::max(mx.y - mn.y, mx.z - mn.z));

    std::vector<BMeshlet> all;
    std::vector<BMeshlet> cur = clusterize(pos, indices, 0, -1.0f, {0, 0, 0});      // :891: LOD 0 carries error -1
    uint32_t lodCount = 1;
    for (uint32_t lod = 0; lod + 1 < kMaxLODCount; lod++) {                         // :895-916
        const float t = (float)lod / (float)kMaxLODCount;
        const float lodErrorAbsolute = (kSimplifyErrorMin + (kSimplifyErrorMax - kSimplifyErrorMin) * t) * scale;
        std::vector<BMeshlet> nextLevel;
        const std::vector<std::vector<uint32_t>> groups = partition_groups(cur);   // MeshletsGMSS :764-880
        for (const auto& g : groups) {
            std::vector<uint32_t> merged;
            for (uint32_t mId : g) for (uint8_t l : cur[mId].tris) merged.push_back(cur[mId].verts[l]);
            float simplificationError = 0.0f;
            const std::vector<uint32_t> simplified = simplify(pos, merged, (size_t)(merged.size() * kGroupSimplifyThreshold), lodErrorAbsolute, &simplificationError);
            if (simplified.empty() || simplified.size() >= (size_t)(merged.size() * kGroupSimplifyMinReduce)) continue;   // :838
            float passedError = 0.0f;
            for (uint32_t mId : g) passedError = std::max(passedError, cur[mId].error);
            const float clusterError = simplificationError + passedError;
            V3 bmn = {FLT_MAX, FLT_MAX, FLT_MAX}, bmx = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
            for (uint32_t v : simplified) { bmn = vmin(bmn, pos[v]); bmx = vmax(bmx, pos[v]); }
            const V3 center = (bmx + bmn) * 0.5f;
            for (uint32_t mId : g) { cur[mId].parentError = clusterError; cur[mId].parentPosCenter = center; }
            std::vector<BMeshlet> made = clusterize(pos, simplified, lod + 1, clusterError, center);
            for (auto& m : made) nextLevel.push_back(std::move(m));
        }
        for (auto& m : cur) all.push_back(std::move(m));
        cur.clear();
        if (nextLevel.empty()) break;
        cur = std::move(nextLevel);
        lodCount++;
    }
    for (auto& m : cur) all.push_back(std::move(m));

    // cluster groups: meshlets with the same (own sphere, parent sphere), at most 4 per group (:330-392)
    std::vector<BGroup> groups;
    std::map<std::array<uint32_t, 8>, uint32_t> open;
    for (uint32_t i = 0; i < all.size(); i++) {
        const BMeshlet& m = all[i];
        std::array<uint32_t, 8> key;
        const float f[8] = {m.clusterPosCenter.x, m.clusterPosCenter.y, m.clusterPosCenter.z, m.error, m.parentPosCenter.x, m.parentPosCenter.y, m.parentPosCenter.z, m.parentError};
        std::memcpy(key.data(), f, sizeof(f));
        auto it = open.find(key);
        if (it == open.end() || groups[it->second].meshlets.size() >= CHORD_GROUP_MAX_MESHLETS) {
            open[key] = (uint32_t)groups.size();
            groups.push_back(BGroup{m.clusterPosCenter, m.error, m.parentPosCenter, m.parentError, {i}});
        } else groups[it->second].meshlets.push_back(i);
    }
    ChordBuiltAsset* a = new ChordBuiltAsset();
    std::vector<uint32_t> order;
    build_tree(groups, a->bvh, order);
    for (uint32_t gi : order) {
        const BGroup& g = groups[gi];
        ChordMeshletGroup o;
        std::memcpy(o.clusterPosCenter, &g.clusterPosCenter, 12); o.error = g.error;
        std::memcpy(o.parentPosCenter, &g.parentPosCenter, 12); o.parentError = g.parentError;
        o.meshletOffset = (uint32_t)a->groupIndices.size(); o.meshletCount = (uint32_t)g.meshlets.size();
        a->groupIndices.insert(a->groupIndices.end(), g.meshlets.begin(), g.meshlets.end());
        a->groups.push_back(o);
    }
    for (const BMeshlet& m : all) {                                                 // asset_gltf_helper.cpp:522-548
        ChordMeshlet o;
        std::memcpy(o.posMin, &m.posMin, 12); std::memcpy(o.posMax, &m.posMax, 12);
        std::memcpy(o.coneAxis, &m.coneAxis, 12); std::memcpy(o.coneApex, &m.coneApex, 12);
        o.coneCutOff = m.coneCutOff; o.lod = m.lod;
        o.dataOffset = (uint32_t)a->meshletData.size();
        o.vertexTriangleCount = ((uint32_t)m.verts.size() & 0xFFu) | ((uint32_t)(m.tris.size() / 3) << 8);
        a->meshletData.insert(a->meshletData.end(), m.verts.begin(), m.verts.end());
        for (size_t t = 0; t < m.tris.size() / 3; t++) a->meshletData.push_back((uint32_t)m.tris[3 * t] | ((uint32_t)m.tris[3 * t + 1] << 8) | ((uint32_t)m.tris[3 * t + 2] << 16));
        a->meshlets.push_back(o);
    }
    a->positions.assign(positionsIn, positionsIn + (size_t)vertexCount * 3);
    if (texcoord0) a->texcoords.assign(texcoord0, texcoord0 + (size_t)vertexCount * 2);
    std::memset(&a->prim, 0, sizeof(a->prim));
    std::memcpy(a->prim.posMin, &mn, 12); std::memcpy(a->prim.posMax, &mx, 12); std::memcpy(a->prim.posAverage, &avg, 12);
    a->prim.vertexCount = vertexCount;
    a->prim.meshletGroupCount = (uint32_t)a->groups.size();
    a->lodCount = lodCount;
    *out = a;
    return CHORDVIS_OK;
}

int chordvis_built_asset_desc(const ChordBuiltAsset* a, ChordAssetDesc* outAsset, ChordPrimitive* outPrimitive, uint32_t* outLodCount)
{
    if (!a || !outAsset || !outPrimitive) return CHORDVIS_E_INVALID;
    std::memset(outAsset, 0, sizeof(*outAsset));
    outAsset->meshlets = a->meshlets.data(); outAsset->meshletCount = (uint32_t)a->meshlets.size();
    outAsset->meshletGroups = a->groups.data(); outAsset->meshletGroupCount = (uint32_t)a->groups.size();
    outAsset->meshletGroupIndices = a->groupIndices.data(); outAsset->meshletGroupIndexCount = (uint32_t)a->groupIndices.size();
    outAsset->meshletData = a->meshletData.data(); outAsset->meshletDataCount = (uint32_t)a->meshletData.size();
    outAsset->positions = a->positions.data(); outAsset->vertexCount = (uint32_t)(a->positions.size() / 3);
    outAsset->texcoord0 = a->texcoords.empty() ? nullptr : a->texcoords.data(); outAsset->texcoord0Count = (uint32_t)(a->texcoords.size() / 2);
    outAsset->bvhNodes = a->bvh.data(); outAsset->bvhNodeCount = (uint32_t)a->bvh.size();
    *outPrimitive = a->prim;
    if (outLodCount) *outLodCount = a->lodCount;
    return CHORDVIS_OK;
}

void chordvis_free_built_asset(ChordBuiltAsset* a) { delete a; }

// A flat little-endian container for a built asset: magic, counts, then the arrays as they stand in memory.  (The
// reference stores cereal binary archives under LZ4, serialize.h:217-320 -- third-party formats of no use without the
// rest of its asset system; this is the minimum that lets a built mesh be kept and reloaded.)
static const char kMagic[8] = {'C', 'H', 'R', 'D', 'A', 'S', '0', '1'};

int chordvis_save_asset(const ChordBuiltAsset* a, const char* path)
{
    if (!a || !path) return CHORDVIS_E_INVALID;
    FILE* f = std::fopen(path, "wb");
    if (!f) return CHORDVIS_E_INVALID;
    const uint64_t counts[8] = {a->positions.size(), a->texcoords.size(), a->meshlets.size(), a->groups.size(), a->groupIndices.size(),
                                a->meshletData.size(), a->bvh.size(), a->lodCount};
    bool ok = std::fwrite(kMagic, 1, 8, f) == 8 && std::fwrite(counts, 8, 8, f) == 8 && std::fwrite(&a->prim, sizeof(a->prim), 1, f) == 1;
    auto put = [&](const void* p, size_t bytes) { if (bytes) ok = ok && std::fwrite(p, 1, bytes, f) == bytes; };
    put(a->positions.data(), a->positions.size() * 4); put(a->texcoords.data(), a->texcoords.size() * 4);
    put(a->meshlets.data(), a->meshlets.size() * sizeof(ChordMeshlet)); put(a->groups.data(), a->groups.size() * sizeof(ChordMeshletGroup));
    put(a->groupIndices.data(), a->groupIndices.size() * 4); put(a->meshletData.data(), a->meshletData.size() * 4);
    put(a->bvh.data(), a->bvh.size() * sizeof(ChordBVHNode));
    ok = std::fclose(f) == 0 && ok;
    return ok ? CHORDVIS_OK : CHORDVIS_E_INVALID;
}

int chordvis_load_asset(const char* path, ChordBuiltAsset** out)
{
    if (!path || !out) return CHORDVIS_E_INVALID;
    *out = nullptr;
    FILE* f = std::fopen(path, "rb");
    if (!f) return CHORDVIS_E_INVALID;
    char magic[8]; uint64_t counts[8];
    ChordBuiltAsset* a = new ChordBuiltAsset();
    bool ok = std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, kMagic, 8) == 0 && std::fread(counts, 8, 8, f) == 8 &&
              std::fread(&a->prim, sizeof(a->prim), 1, f) == 1;
    for (int i = 0; ok && i < 7; i++) ok = counts[i] < (1ull << 32);
    auto get = [&](auto& vec, uint64_t n) { if (!ok) return; vec.resize((size_t)n); if (n) ok = std::fread(vec.data(), sizeof(vec[0]), (size_t)n, f) == n; };
    if (ok) { get(a->positions, counts[0]); get(a->texcoords, counts[1]); get(a->meshlets, counts[2]); get(a->groups, counts[3]);
              get(a->groupIndices, counts[4]); get(a->meshletData, counts[5]); get(a->bvh, counts[6]); a->lodCount = (uint32_t)counts[7]; }
    std::fclose(f);
    if (!ok) { delete a; return CHORDVIS_E_INVALID; }
    *out = a;
    return CHORDVIS_OK;
}

} // extern "C"

// ---- the reference's own container for the geometry of an asset: GLTFBinary (asset_gltf.h:260-300) --------------------------
// saveAsset / loadAsset (serialize.h:217-320): a cereal BinaryOutputArchive of {AssetCompressedMeta, std::string}, the string
// being the cereal binary archive of the GLTFBinary, LZ4-block-compressed or not.  Both formats are public and small, and both
// are restated here from their specifications (no cereal, no liblz4 in this library):
//   cereal binary: arithmetic values little-endian as they stand; std::string / std::vector = uint64 count + elements; a class
//     registered with CEREAL_CLASS_VERSION writes its uint32 version ONCE per archive, in front of the first instance's members;
//     members in the order the serialize function names them (serialize.h:47-99: NOT the memory order for GLTFMeshlet and
//     GLTFBVHNode); glm vectors component by component (pch.h:100-121); enum class as size_t (utils.h:115-118).
//   LZ4 block: sequences of {token, [literal length bytes], literals, offset16, [match length bytes]}, min match 4, the last
//     sequence literals only.
// Pinned by tests/golden/gltf_binary_{raw,lz4}.bin, which the reference's vendored cereal and LZ4 wrote
// (tests/golden/make_gltf_binary_fixture.{cpp,sh}).  Vertex attributes this path never reads (normals, tangents, second UV set,
// colours, smooth normals, LOD-0 indices) are skipped on load and written empty.

namespace {

bool lz4_block_decode(const uint8_t* src, size_t n, uint8_t* dst, size_t cap)
{
    size_t i = 0, o = 0;
    while (i < n) {
        const uint32_t token = src[i++];
        size_t lit = token >> 4;
        if (lit == 15) { uint8_t b; do { if (i >= n) return false; b = src[i++]; lit += b; } while (b == 255); }
        if (i + lit > n || o + lit > cap) return false;
        std::memcpy(dst + o, src + i, lit); i += lit; o += lit;
        if (i >= n) break;                                         // the last sequence has no match
        if (i + 2 > n) return false;
        const size_t off = (size_t)src[i] | ((size_t)src[i + 1] << 8); i += 2;
        if (off == 0 || off > o) return false;
        size_t len = token & 15u;
        if (len == 15) { uint8_t b; do { if (i >= n) return false; b = src[i++]; len += b; } while (b == 255); }
        len += 4;
        if (o + len > cap) return false;
        for (size_t k = 0; k < len; k++) dst[o + k] = dst[o + k - off];   // (may overlap: byte by byte)
        o += len;
    }
    return o == cap;
}

// greedy single-probe matcher (a 4-byte hash of the position, 64 KB window): any valid LZ4 block will do
void lz4_block_encode(const uint8_t* src, size_t n, std::vector<uint8_t>& out)
{
    out.clear();
    std::vector<uint32_t> table(1u << 16, 0xFFFFFFFFu);
    auto put_len = [&](size_t v) { while (v >= 255) { out.push_back(255); v -= 255; } out.push_back((uint8_t)v); };
    auto emit = [&](size_t litStart, size_t litLen, size_t off, size_t matchLen) {      // matchLen 0: last sequence
        const size_t ml = matchLen ? matchLen - 4 : 0;
        out.push_back((uint8_t)(((litLen < 15 ? litLen : 15) << 4) | (matchLen ? (ml < 15 ? ml : 15) : 0)));
        if (litLen >= 15) put_len(litLen - 15);
        out.insert(out.end(), src + litStart, src + litStart + litLen);
        if (matchLen) { out.push_back((uint8_t)(off & 255)); out.push_back((uint8_t)(off >> 8)); if (ml >= 15) put_len(ml - 15); }
    };
    size_t anchor = 0, i = 0;
    // (format rules: the last 5 bytes are literals, the last match starts at least 12 bytes before the end)
    const size_t matchLimit = n > 12 ? n - 12 : 0, endLimit = n > 5 ? n - 5 : 0;
    while (i < matchLimit) {
        uint32_t v; std::memcpy(&v, src + i, 4);
        const uint32_t h = (v * 2654435761u) >> 16;
        const uint32_t cand = table[h];
        table[h] = (uint32_t)i;
        uint32_t w = 0;
        if (cand != 0xFFFFFFFFu && i - cand <= 65535) std::memcpy(&w, src + cand, 4);
        if (cand == 0xFFFFFFFFu || i - cand > 65535 || w != v) { i++; continue; }
        size_t len = 4;
        while (i + len < endLimit && src[cand + len] == src[i + len]) len++;
        emit(anchor, i - anchor, i - cand, len);
        i += len; anchor = i;
    }
    emit(anchor, n - anchor, 0, 0);
}

struct ByteReader {
    const uint8_t* p; size_t n, i = 0; bool ok = true;
    template <class T> T get() { T v{}; if (i + sizeof(T) > n) { ok = false; return v; } std::memcpy(&v, p + i, sizeof(T)); i += sizeof(T); return v; }
    void skip(uint64_t bytes) { if (bytes > n - i) ok = false; else i += (size_t)bytes; }
    void floats(std::vector<float>& out, uint64_t count) { if (count > (n - i) / 4) { ok = false; return; } out.resize((size_t)count); if (count) std::memcpy(out.data(), p + i, (size_t)count * 4); i += (size_t)count * 4; }
};
struct ByteWriter {
    std::vector<uint8_t> b;
    template <class T> void put(const T& v) { const uint8_t* q = reinterpret_cast<const uint8_t*>(&v); b.insert(b.end(), q, q + sizeof(T)); }
    void raw(const void* q, size_t bytes) { const uint8_t* c = static_cast<const uint8_t*>(q); b.insert(b.end(), c, c + bytes); }
};

} // namespace

extern "C" {

// The reference's GLTFBinary archive of a built asset (compressionMode None / Lz4).
int chordvis_save_gltf_binary(const ChordBuiltAsset* a, const char* path, int lz4)
{
    if (!a || !path) return CHORDVIS_E_INVALID;
    ByteWriter w;
    w.put<uint32_t>(0u);                                                             // GLTFBinary: class version (kAssetVersion = 0)
    const uint64_t nv = a->positions.size() / 3;
    w.put<uint64_t>(nv); w.raw(a->positions.data(), a->positions.size() * 4);       // positions (vec3 by component = the floats as they stand)
    w.put<uint64_t>(0);                                                              // normals
    w.put<uint64_t>(a->texcoords.size() / 2); w.raw(a->texcoords.data(), a->texcoords.size() * 4);   // texcoords0
    w.put<uint64_t>(0);                                                              // tangents
    w.put<uint64_t>(0); w.put<uint64_t>(0); w.put<uint64_t>(0);                     // smoothNormals, texcoords1, colors0
    w.put<uint64_t>(a->meshlets.size());
    for (size_t i = 0; i < a->meshlets.size(); i++) {                                // serialize.h:64-75 member order
        const ChordMeshlet& m = a->meshlets[i];
        if (i == 0) w.put<uint32_t>(0u);
        w.raw(m.posMin, 12); w.put(m.dataOffset); w.raw(m.posMax, 12); w.put(m.vertexTriangleCount);
        w.put(m.coneCutOff); w.raw(m.coneAxis, 12); w.raw(m.coneApex, 12); w.put(m.lod);
    }
    w.put<uint64_t>(a->meshletData.size()); w.raw(a->meshletData.data(), a->meshletData.size() * 4);
    w.put<uint64_t>(a->bvh.size());
    for (size_t i = 0; i < a->bvh.size(); i++) {                                     // serialize.h:47-54
        const ChordBVHNode& n = a->bvh[i];
        if (i == 0) w.put<uint32_t>(0u);
        w.raw(n.sphere, 16); w.raw(n.children, 32); w.put(n.leafMeshletGroupOffset); w.put(n.leafMeshletGroupCount); w.put(n.bvhNodeCount);
    }
    w.put<uint64_t>(a->groups.size());
    for (size_t i = 0; i < a->groups.size(); i++) {                                  // serialize.h:55-63 (= memory order)
        if (i == 0) w.put<uint32_t>(0u);
        w.raw(&a->groups[i], sizeof(ChordMeshletGroup));
    }
    w.put<uint64_t>(a->groupIndices.size()); w.raw(a->groupIndices.data(), a->groupIndices.size() * 4);
    w.put<uint64_t>(0);                                                              // lod0Indices
    if (w.b.size() > 0x7FFFFFFFull) return CHORDVIS_E_INVALID;                       // (AssetCompressedMeta holds int32 sizes)
    std::vector<uint8_t> packed;
    if (lz4) lz4_block_encode(w.b.data(), w.b.size(), packed);
    const std::vector<uint8_t>& body = lz4 ? packed : w.b;
    ByteWriter f;
    f.put<int32_t>((int32_t)w.b.size()); f.put<int32_t>((int32_t)body.size()); f.put<uint64_t>(lz4 ? 1u : 0u);   // meta (serialize.h:209-213)
    f.put<uint64_t>(body.size()); f.raw(body.data(), body.size());                                                // std::string
    FILE* fp = std::fopen(path, "wb");
    if (!fp) return CHORDVIS_E_INVALID;
    bool ok = std::fwrite(f.b.data(), 1, f.b.size(), fp) == f.b.size();
    ok = std::fclose(fp) == 0 && ok;
    return ok ? CHORDVIS_OK : CHORDVIS_E_INVALID;
}

// Reads a GLTFBinary archive into a built asset holding ONE primitive that spans the whole file (the reference keeps the
// per-primitive offsets in its GLTFAsset, a different archive: a host that has them fills its own ChordPrimitive records and
// uses only the arrays of chordvis_built_asset_desc).
static int load_gltf_binary_impl(const char* path, ChordBuiltAsset** out);

// (the file's header sizes are not trusted: every element count is checked against the bytes that remain BEFORE it is multiplied,
// the decompressed size against what an LZ4 block of that length can expand to; allocation failures do not cross the C boundary)
int chordvis_load_gltf_binary(const char* path, ChordBuiltAsset** out)
{
    try { return load_gltf_binary_impl(path, out); }
    catch (const std::exception&) { if (out) *out = nullptr; return CHORDVIS_E_INVALID; }
}

static int load_gltf_binary_impl(const char* path, ChordBuiltAsset** out)
{
    if (!path || !out) return CHORDVIS_E_INVALID;
    *out = nullptr;
    FILE* fp = std::fopen(path, "rb");
    if (!fp) return CHORDVIS_E_INVALID;
    std::vector<uint8_t> file;
    { uint8_t buf[65536]; size_t k; while ((k = std::fread(buf, 1, sizeof(buf), fp)) > 0) file.insert(file.end(), buf, buf + k); }
    std::fclose(fp);
    ByteReader f{file.data(), file.size()};
    const int32_t rawSize = f.get<int32_t>(), compSize = f.get<int32_t>();
    const uint64_t mode = f.get<uint64_t>(), strLen = f.get<uint64_t>();
    if (!f.ok || rawSize < 0 || compSize < 0 || strLen != (uint64_t)compSize || strLen > file.size() - f.i || mode > 1) return CHORDVIS_E_INVALID;
    if (mode == 1 && (uint64_t)rawSize > (uint64_t)compSize * 255u + 64u) return CHORDVIS_E_INVALID;   // (an LZ4 block expands at most 255 : 1)
    std::vector<uint8_t> raw;
    if (mode == 1) { raw.resize((size_t)rawSize); if (!lz4_block_decode(file.data() + f.i, (size_t)compSize, raw.data(), raw.size())) return CHORDVIS_E_INVALID; }
    else { if (compSize != rawSize) return CHORDVIS_E_INVALID; raw.assign(file.begin() + (long)f.i, file.begin() + (long)(f.i + strLen)); }
    ByteReader r{raw.data(), raw.size()};
    std::unique_ptr<ChordBuiltAsset> holder(new ChordBuiltAsset());     // (freed on every early return and on bad_alloc)
    ChordBuiltAsset* a = holder.get();
    (void)r.get<uint32_t>();                                                         // GLTFBinary class version
    // an element count times its size, refused (reader marked bad) when the product exceeds the bytes that remain
    auto bytes_of = [&](uint64_t elem) -> uint64_t { const uint64_t n = r.get<uint64_t>(); if (n > (raw.size() - r.i) / elem) { r.ok = false; return 0; } return n * elem; };
    r.floats(a->positions, bytes_of(12) / 4);
    r.skip(bytes_of(12));                                                            // normals
    r.floats(a->texcoords, bytes_of(8) / 4);
    r.skip(bytes_of(16));                                                            // tangents
    r.skip(bytes_of(12)); r.skip(bytes_of(8)); r.skip(bytes_of(16));                 // smoothNormals, texcoords1, colors0
    uint64_t n = r.get<uint64_t>();
    if (n > (raw.size() - r.i) / 64) r.ok = false;
    for (uint64_t i = 0; r.ok && i < n; i++) {
        if (i == 0) (void)r.get<uint32_t>();
        ChordMeshlet m;
        for (int k = 0; k < 3; k++) m.posMin[k] = r.get<float>();
        m.dataOffset = r.get<uint32_t>();
        for (int k = 0; k < 3; k++) m.posMax[k] = r.get<float>();
        m.vertexTriangleCount = r.get<uint32_t>(); m.coneCutOff = r.get<float>();
        for (int k = 0; k < 3; k++) m.coneAxis[k] = r.get<float>();
        for (int k = 0; k < 3; k++) m.coneApex[k] = r.get<float>();
        m.lod = r.get<uint32_t>();
        a->meshlets.push_back(m);
    }
    n = r.get<uint64_t>();
    if (n > (raw.size() - r.i) / 4) r.ok = false; else { a->meshletData.resize((size_t)n); for (uint64_t i = 0; i < n; i++) a->meshletData[(size_t)i] = r.get<uint32_t>(); }
    n = r.get<uint64_t>();
    if (n > (raw.size() - r.i) / 60) r.ok = false;
    for (uint64_t i = 0; r.ok && i < n; i++) {
        if (i == 0) (void)r.get<uint32_t>();
        ChordBVHNode b;
        for (int k = 0; k < 4; k++) b.sphere[k] = r.get<float>();
        for (int k = 0; k < 8; k++) b.children[k] = r.get<uint32_t>();
        b.leafMeshletGroupOffset = r.get<uint32_t>(); b.leafMeshletGroupCount = r.get<uint32_t>(); b.bvhNodeCount = r.get<uint32_t>();
        a->bvh.push_back(b);
    }
    n = r.get<uint64_t>();
    if (n > (raw.size() - r.i) / 40) r.ok = false;
    for (uint64_t i = 0; r.ok && i < n; i++) {
        if (i == 0) (void)r.get<uint32_t>();
        a->groups.push_back(r.get<ChordMeshletGroup>());
    }
    n = r.get<uint64_t>();
    if (n > (raw.size() - r.i) / 4) r.ok = false; else { a->groupIndices.resize((size_t)n); for (uint64_t i = 0; i < n; i++) a->groupIndices[(size_t)i] = r.get<uint32_t>(); }
    r.skip(bytes_of(4));                                                             // lod0Indices
    if (!r.ok || r.i != raw.size() || a->positions.empty()) return CHORDVIS_E_INVALID;
    if (!a->texcoords.empty() && a->texcoords.size() / 2 != a->positions.size() / 3) return CHORDVIS_E_INVALID;   // one uv per vertex, or none
    // one primitive over everything
    std::memset(&a->prim, 0, sizeof(a->prim));
    const size_t nv = a->positions.size() / 3;
    float mn[3] = {a->positions[0], a->positions[1], a->positions[2]}, mx[3] = {mn[0], mn[1], mn[2]};
    double sum[3] = {0, 0, 0};
    for (size_t v = 0; v < nv; v++) for (int k = 0; k < 3; k++) { const float x = a->positions[3 * v + k]; mn[k] = std::min(mn[k], x); mx[k] = std::max(mx[k], x); sum[k] += x; }
    for (int k = 0; k < 3; k++) { a->prim.posMin[k] = mn[k]; a->prim.posMax[k] = mx[k]; a->prim.posAverage[k] = (float)(sum[k] / (double)nv); }
    a->prim.vertexCount = (uint32_t)nv;
    a->prim.meshletGroupCount = (uint32_t)a->groups.size();
    uint32_t lods = 0;
    for (const ChordMeshlet& m : a->meshlets) lods = std::max(lods, m.lod + 1u);
    a->lodCount = lods;
    *out = holder.release();
    return CHORDVIS_OK;
}

} // extern "C"
