// Host-side view / object record construction for the visibility path.
//
// Mirrors the reference's CPU code that feeds the path (no device work):
//   ICamera::fillViewUniformParameter        source/utils/camera.cpp:17-78
//   ICamera::computeRelativeWorldFrustum     source/utils/camera.cpp:80-154
//   infiniteInvertZPerspectiveRH_ZO          source/utils/utils.cpp:186-198
//   ViewportCamera::updateMatrixMisc         application/flower/widget/viewport.cpp:434-445
//   main-view InstanceCullingViewInfo fill   source/renderer/renderer.cpp:251-263
//   SceneNode::getObjectBasicData            source/scene/scene_node.cpp:42-90
//   buildHZB extent math                     source/renderer/postprocessing/hzb.cpp:49-63
//
// The reference does this with glm (column-major, float for view data, double
// for world transforms); this file restates the same formulas without glm.

#include "../../include/chordvis.h"

#include <cmath>
#include <vector>
#include <cstring>

namespace {

struct V3 { float x, y, z; };

inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 mul(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline V3 normalize(V3 a) { float l = std::sqrt(dot(a, a)); return {a.x / l, a.y / l, a.z / l}; }

inline float& at(ChordMat4& M, int r, int c) { return M.m[c * 4 + r]; }
inline float at(const ChordMat4& M, int r, int c) { return M.m[c * 4 + r]; }

ChordMat4 matmul(const ChordMat4& A, const ChordMat4& B)
{
    ChordMat4 C;
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++)
            at(C, r, c) = ((at(A, r, 0) * at(B, 0, c) + at(A, r, 1) * at(B, 1, c)) + at(A, r, 2) * at(B, 2, c)) + at(A, r, 3) * at(B, 3, c);
    return C;
}

// General 4x4 inverse by cofactors (what glm::inverse computes), templated for f32/f64.
template <typename T>
bool inverse4(const T* m /*column-major*/, T* out)
{
    T inv[16];
    inv[0]  =  m[5]*m[10]*m[15] - m[5]*m[11]*m[14] - m[9]*m[6]*m[15] + m[9]*m[7]*m[14] + m[13]*m[6]*m[11] - m[13]*m[7]*m[10];
    inv[4]  = -m[4]*m[10]*m[15] + m[4]*m[11]*m[14] + m[8]*m[6]*m[15] - m[8]*m[7]*m[14] - m[12]*m[6]*m[11] + m[12]*m[7]*m[10];
    inv[8]  =  m[4]*m[9]*m[15]  - m[4]*m[11]*m[13] - m[8]*m[5]*m[15] + m[8]*m[7]*m[13] + m[12]*m[5]*m[11] - m[12]*m[7]*m[9];
    inv[12] = -m[4]*m[9]*m[14]  + m[4]*m[10]*m[13] + m[8]*m[5]*m[14] - m[8]*m[6]*m[13] - m[12]*m[5]*m[10] + m[12]*m[6]*m[9];
    inv[1]  = -m[1]*m[10]*m[15] + m[1]*m[11]*m[14] + m[9]*m[2]*m[15] - m[9]*m[3]*m[14] - m[13]*m[2]*m[11] + m[13]*m[3]*m[10];
    inv[5]  =  m[0]*m[10]*m[15] - m[0]*m[11]*m[14] - m[8]*m[2]*m[15] + m[8]*m[3]*m[14] + m[12]*m[2]*m[11] - m[12]*m[3]*m[10];
    inv[9]  = -m[0]*m[9]*m[15]  + m[0]*m[11]*m[13] + m[8]*m[1]*m[15] - m[8]*m[3]*m[13] - m[12]*m[1]*m[11] + m[12]*m[3]*m[9];
    inv[13] =  m[0]*m[9]*m[14]  - m[0]*m[10]*m[13] - m[8]*m[1]*m[14] + m[8]*m[2]*m[13] + m[12]*m[1]*m[10] - m[12]*m[2]*m[9];
    inv[2]  =  m[1]*m[6]*m[15]  - m[1]*m[7]*m[14]  - m[5]*m[2]*m[15] + m[5]*m[3]*m[14] + m[13]*m[2]*m[7]  - m[13]*m[3]*m[6];
    inv[6]  = -m[0]*m[6]*m[15]  + m[0]*m[7]*m[14]  + m[4]*m[2]*m[15] - m[4]*m[3]*m[14] - m[12]*m[2]*m[7]  + m[12]*m[3]*m[6];
    inv[10] =  m[0]*m[5]*m[15]  - m[0]*m[7]*m[13]  - m[4]*m[1]*m[15] + m[4]*m[3]*m[13] + m[12]*m[1]*m[7]  - m[12]*m[3]*m[5];
    inv[14] = -m[0]*m[5]*m[14]  + m[0]*m[6]*m[13]  + m[4]*m[1]*m[14] - m[4]*m[2]*m[13] - m[12]*m[1]*m[6]  + m[12]*m[2]*m[5];
    inv[3]  = -m[1]*m[6]*m[11]  + m[1]*m[7]*m[10]  + m[5]*m[2]*m[11] - m[5]*m[3]*m[10] - m[9]*m[2]*m[7]   + m[9]*m[3]*m[6];
    inv[7]  =  m[0]*m[6]*m[11]  - m[0]*m[7]*m[10]  - m[4]*m[2]*m[11] + m[4]*m[3]*m[10] + m[8]*m[2]*m[7]   - m[8]*m[3]*m[6];
    inv[11] = -m[0]*m[5]*m[11]  + m[0]*m[7]*m[9]   + m[4]*m[1]*m[11] - m[4]*m[3]*m[9]  - m[8]*m[1]*m[7]   + m[8]*m[3]*m[5];
    inv[15] =  m[0]*m[5]*m[10]  - m[0]*m[6]*m[9]   - m[4]*m[1]*m[10] + m[4]*m[2]*m[9]  + m[8]*m[1]*m[6]   - m[8]*m[2]*m[5];
    T det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    if (det == T(0)) return false;
    T inv_det = T(1) / det;
    for (int i = 0; i < 16; i++) out[i] = inv[i] * inv_det;
    return true;
}

inline uint32_t nextPOT(uint32_t v) { v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; v++; return v; }

} // namespace

extern "C" {

const char* chordvis_version(void) { return "chordvis-mi355x 0.1 (gfx950)"; }

int chordvis_hzb_desc(uint32_t srcWidth, uint32_t srcHeight, ChordHZBDesc* out)
{
    if (!out || srcWidth < 2 || srcHeight < 2 || srcWidth > 8192 || srcHeight > 8192) return CHORDVIS_E_INVALID;
    uint32_t w = nextPOT(srcWidth) / 2, h = nextPOT(srcHeight) / 2;     // hzb.cpp:52-53
    if (w == srcWidth) w /= 2;                                          // :55
    if (h == srcHeight) h /= 2;                                         // :56
    if (w < 1) w = 1;
    if (h < 1) h = 1;
    uint32_t mx = w > h ? w : h, mips = 0;
    while (mx) { mips++; mx >>= 1; }                                    // getMipLevelsCount, utils.h:500-503
    if (mips > CHORD_HZB_MAX_MIPS) mips = CHORD_HZB_MAX_MIPS;
    std::memset(out, 0, sizeof(*out));
    out->srcWidth = srcWidth; out->srcHeight = srcHeight;
    out->width = w; out->height = h; out->mipCount = mips;
    uint32_t off = 0;
    for (uint32_t l = 0; l < mips; l++) {
        uint32_t mw = w >> l, mh = h >> l;
        if (mw < 1) mw = 1;
        if (mh < 1) mh = 1;
        out->mipOffset[l] = off;
        off += mw * mh;
    }
    out->totalTexels = off;
    return CHORDVIS_OK;
}

int chordvis_camera_fill_view(const ChordCameraDesc* cam, const ChordCameraView* lastFrame,
                              ChordCameraView* outView, ChordInstanceCullingView* outIV)
{
    if (!cam || !outView || !outIV || cam->width == 0 || cam->height == 0) return CHORDVIS_E_INVALID;

    // viewport.cpp:267-268 — basis from front + world up (double), then cast to float (viewport.cpp:438)
    const double fx = cam->front[0], fy = cam->front[1], fz = cam->front[2];
    const double ux = cam->worldUp[0], uy = cam->worldUp[1], uz = cam->worldUp[2];
    double rx = fy * uz - uy * fz, ry = fz * ux - uz * fx, rz = fx * uy - ux * fy;
    double rl = std::sqrt(rx * rx + ry * ry + rz * rz);
    if (rl == 0.0) return CHORDVIS_E_INVALID;
    rx /= rl; ry /= rl; rz /= rl;
    double upx = ry * fz - fy * rz, upy = rz * fx - fz * rx, upz = rx * fy - fx * ry;
    double ul = std::sqrt(upx * upx + upy * upy + upz * upz);
    upx /= ul; upy /= ul; upz /= ul;

    const V3 front = {(float)fx, (float)fy, (float)fz};
    const V3 up = {(float)upx, (float)upy, (float)upz};
    const V3 right = {(float)rx, (float)ry, (float)rz};

    // glm::lookAtRH(eye = 0, center = front, up)
    const V3 f = normalize(front);
    const V3 s = normalize(cross(f, up));
    const V3 u = cross(s, f);
    ChordMat4 view;
    std::memset(&view, 0, sizeof(view));
    at(view, 0, 0) = s.x;  at(view, 0, 1) = s.y;  at(view, 0, 2) = s.z;
    at(view, 1, 0) = u.x;  at(view, 1, 1) = u.y;  at(view, 1, 2) = u.z;
    at(view, 2, 0) = -f.x; at(view, 2, 1) = -f.y; at(view, 2, 2) = -f.z;
    at(view, 3, 3) = 1.0f;

    // utils.cpp:186-198
    const float aspect = (float)cam->width / (float)cam->height;
    const float tanHalfFovy = std::tan(cam->fovy * 0.5f);
    ChordMat4 proj;
    std::memset(&proj, 0, sizeof(proj));
    at(proj, 0, 0) = 1.0f / (aspect * tanHalfFovy);
    at(proj, 1, 1) = 1.0f / tanHalfFovy;
    at(proj, 3, 2) = -1.0f;               // glm result[2][3]
    at(proj, 2, 3) = (float)cam->zNear;   // glm result[3][2]

    // camera.cpp:36-51 — jitter matrix (translation of clip x,y) * projection
    ChordMat4 jit;
    std::memset(&jit, 0, sizeof(jit));
    at(jit, 0, 0) = at(jit, 1, 1) = at(jit, 2, 2) = at(jit, 3, 3) = 1.0f;
    at(jit, 0, 3) += 2.0f * cam->jitter[0] / (float)cam->width;
    at(jit, 1, 3) += -2.0f * cam->jitter[1] / (float)cam->height;
    const ChordMat4 projection = matmul(jit, proj);

    std::memset(outView, 0, sizeof(*outView));
    outView->translatedWorldToView = view;
    outView->translatedWorldToClip = matmul(projection, view);
    outView->translatedWorldToClipLastFrame = lastFrame ? lastFrame->translatedWorldToClip : outView->translatedWorldToClip; // renderer.cpp:201
    outView->renderDimension[0] = 1.0f * (float)cam->width;             // renderer.cpp:193-198
    outView->renderDimension[1] = 1.0f * (float)cam->height;
    outView->renderDimension[2] = 1.0f / (float)cam->width;
    outView->renderDimension[3] = 1.0f / (float)cam->height;
    outView->cameraFovy = cam->fovy;
    outView->zNear = (float)cam->zNear;
    outView->zFar = (float)cam->zFar;
    {   // camera.cpp:25-31 with viewport.cpp:444: glm::perspectiveRH_ZO(fovy, aspect, zNear := zFar, zFar := zNear) -- still reverse Z
        const float zN = (float)cam->zFar, zF = (float)cam->zNear;
        ChordMat4 pz;
        std::memset(&pz, 0, sizeof(pz));
        at(pz, 0, 0) = 1.0f / (aspect * tanHalfFovy);
        at(pz, 1, 1) = 1.0f / tanHalfFovy;
        at(pz, 2, 2) = zF / (zN - zF);
        at(pz, 3, 2) = -1.0f;
        at(pz, 2, 3) = -(zF * zN) / (zF - zN);
        const ChordMat4 vpz = matmul(pz, view);
        if (!inverse4<float>(vpz.m, outView->clipToTranslatedWorldWithZFar_NoJitter.m))
            std::memset(&outView->clipToTranslatedWorldWithZFar_NoJitter, 0, sizeof(ChordMat4));
    }
    // projectSphereToScreen's view-constant factor, base.hlsli:503-518: height * 0.5 / tan(fovy / 2)
    outView->lodScale = outView->renderDimension[1] * 0.5f / std::tan(0.5f * cam->fovy);

    // camera.cpp:80-154
    float planes[6][4];
    {
        const V3 camWorldPos = {0.0f, 0.0f, 0.0f};
        const V3 forwardVector = normalize(front);
        const V3 upVector = normalize(up);
        const V3 rightVector = normalize(right);
        const V3 nearC = add(camWorldPos, mul(forwardVector, (float)cam->zNear));
        const V3 farC = add(camWorldPos, mul(forwardVector, (float)cam->zFar));
        const float yNearHalf = (float)(cam->zNear * (double)tanHalfFovy);
        const float yFarHalf = (float)(cam->zFar * (double)tanHalfFovy);
        const V3 yNearHalfV = mul(upVector, yNearHalf);
        const V3 xNearHalfV = mul(rightVector, yNearHalf * aspect);
        const V3 yFarHalfV = mul(upVector, yFarHalf);
        const V3 xFarHalfV = mul(rightVector, yFarHalf * aspect);
        const V3 nrt = add(add(nearC, xNearHalfV), yNearHalfV);
        const V3 nrd = sub(add(nearC, xNearHalfV), yNearHalfV);
        const V3 nlt = add(sub(nearC, xNearHalfV), yNearHalfV);
        const V3 nld = sub(sub(nearC, xNearHalfV), yNearHalfV);
        const V3 frt = add(add(farC, xFarHalfV), yFarHalfV);
        const V3 frd = sub(add(farC, xFarHalfV), yFarHalfV);
        const V3 flt = add(sub(farC, xFarHalfV), yFarHalfV);
        const V3 fld = sub(sub(farC, xFarHalfV), yFarHalfV);
        auto plane = [&](int i, V3 pC, V3 p1, V3 p2) {
            const V3 n = normalize(cross(sub(p1, pC), sub(p2, pC)));
            planes[i][0] = n.x; planes[i][1] = n.y; planes[i][2] = n.z; planes[i][3] = -dot(n, pC);
        };
        plane(0, fld, flt, nld);   // left
        plane(1, frd, fld, nrd);   // down
        plane(2, frt, frd, nrt);   // right
        plane(3, flt, frt, nlt);   // top
        plane(4, nrt, nrd, nlt);   // front
        plane(5, frt, flt, frd);   // back
    }

    // renderer.cpp:251-263
    std::memset(outIV, 0, sizeof(*outIV));
    std::memcpy(outIV->frustumPlanesRS, planes, sizeof(planes));
    outIV->translatedWorldToClip = outView->translatedWorldToClip;
    if (!inverse4<float>(outIV->translatedWorldToClip.m, outIV->clipToTranslatedWorld.m))
        std::memset(&outIV->clipToTranslatedWorld, 0, sizeof(ChordMat4));
    for (int i = 0; i < 3; i++) std::memcpy(&outIV->cameraWorldPos[i * 2], &cam->position[i], 8);   // fillDouble3, camera.cpp:6-15
    std::memcpy(outIV->renderDimension, outView->renderDimension, sizeof(float) * 4);
    return CHORDVIS_OK;
}

int chordvis_object_basic_data(const double localToWorld[16], const double prevLocalToWorld[16],
                               const double cameraPos[3], const double cameraPosLast[3],
                               ChordObjectBasicData* out)
{
    if (!localToWorld || !cameraPos || !out) return CHORDVIS_E_INVALID;
    const double* prev = prevLocalToWorld ? prevLocalToWorld : localToWorld;
    const double* camLast = cameraPosLast ? cameraPosLast : cameraPos;
    std::memset(out, 0, sizeof(*out));
    double m[16], p[16];
    std::memcpy(m, localToWorld, sizeof(m));
    std::memcpy(p, prev, sizeof(p));
    m[12] -= cameraPos[0]; m[13] -= cameraPos[1]; m[14] -= cameraPos[2];       // scene_node.cpp:51-55
    p[12] -= camLast[0];   p[13] -= camLast[1];   p[14] -= camLast[2];         // :80-84
    for (int i = 0; i < 16; i++) {
        out->localToTranslatedWorld.m[i] = (float)m[i];
        out->localToTranslatedWorldLastFrame.m[i] = (float)p[i];
    }
    if (!inverse4<float>(out->localToTranslatedWorld.m, out->translatedWorldToLocal.m)) return CHORDVIS_E_INVALID;
    const float* L = out->localToTranslatedWorld.m;
    float sx = std::sqrt(L[0] * L[0] + L[1] * L[1] + L[2] * L[2]);             // :60-64
    float sy = std::sqrt(L[4] * L[4] + L[5] * L[5] + L[6] * L[6]);
    float sz = std::sqrt(L[8] * L[8] + L[9] * L[9] + L[10] * L[10]);
    float mx = std::fmax(std::fmax(std::fabs(sx), std::fabs(sy)), std::fabs(sz));
    out->scaleExtractFromMatrix[0] = sx; out->scaleExtractFromMatrix[1] = sy;
    out->scaleExtractFromMatrix[2] = sz; out->scaleExtractFromMatrix[3] = mx;
    return CHORDVIS_OK;
}

int chordvis_object_basic_data_batch(uint32_t count, const double* localToWorld, const double* prevLocalToWorld,
                                     const double cameraPos[3], const double cameraPosLast[3], ChordObject* objects)
{
    if (!localToWorld || !objects) return CHORDVIS_E_INVALID;
    for (uint32_t i = 0; i < count; i++) {
        int rc = chordvis_object_basic_data(localToWorld + (size_t)i * 16,
                                            prevLocalToWorld ? prevLocalToWorld + (size_t)i * 16 : nullptr,
                                            cameraPos, cameraPosLast, &objects[i].basicData);
        if (rc != CHORDVIS_OK) return rc;
    }
    return CHORDVIS_OK;
}

// cascadeComputeCS (cascade_setup.hlsl:79-372) on the host: a TRANSLITERATION of the shader, statement for statement and with
// its identifiers (splitLambda, splitCascadeCount, stableDistance, zStartBiasScale, radiusScale, the argument order at :163, the
// reuse of frontN at :365) -- it has to produce the same fp32 values, so the operation order is the shader's and nothing here
// is an independent design.  The reference runs these few hundred flops per cascade as a
// one-group compute shader only because it reads the SDSM depth range from a GPU buffer; here the caller hands that range
// over (or NULL: the shader's own "no valid range buffer" branch, :118).  fp32 in the shader's operation order.
int chordvis_cascade_setup(const ChordCascadeConfig* cfg, const ChordCameraView* view, const ChordInstanceCullingView* mainInstanceView,
                           const float lightDirIn[3], const uint32_t validDepthMinMax[2], uint32_t tickCount, int bCacheValid,
                           ChordInstanceCullingView* views)
{
    if (!cfg || !view || !mainInstanceView || !lightDirIn || !views) return CHORDVIS_E_INVALID;
    if (cfg->cascadeCount < 1 || (uint32_t)cfg->cascadeCount > CHORD_MAX_CASCADES || cfg->realtimeCascadeCount < 0 ||
        cfg->realtimeCascadeCount > cfg->cascadeCount || cfg->cascadeDim < 64) return CHORDVIS_E_INVALID;
    const uint32_t cascadeCount = (uint32_t)cfg->cascadeCount, realtimeCount = (uint32_t)cfg->realtimeCascadeCount;
    const V3 upDir = {0.0f, 1.0f, 0.0f};
    const float nearZ = view->zNear, farZ = view->zFar, clipRange = farZ - nearZ;
    auto logCascadeSplit = [&](float farDepthPlane, float nearDepthPlane, uint32_t cascadeId, uint32_t count, float lambda) {
        const float range = farDepthPlane - nearDepthPlane, ratio = farDepthPlane / nearDepthPlane;      // :56-74
        const float p = (float)(cascadeId + 1) / (float)count;
        // (pow is the one transcendental of the shader; pinned as the binary64 result rounded once -- tests/spec_np.py
        // cascade_views_f32 and tests/golden/cascade_setup.json hold this function to the bit)
        const float logScale = nearDepthPlane * (float)std::pow((double)std::fabs(ratio), (double)p);
        const float uniformScale = nearDepthPlane + range * p;
        const float d = lambda * (logScale - uniformScale) + uniformScale;
        return (d - nearZ) / clipRange;
    };
    auto mulPoint = [](const ChordMat4& M, float x, float y, float z, float w, float out[4]) {
        for (int r = 0; r < 4; r++) out[r] = ((at(M, r, 0) * x + at(M, r, 1) * y) + at(M, r, 2) * z) + at(M, r, 3) * w;
    };
    struct Cascade { float sphereRadius, cascadeSphereRadius, sphereRadius0, minZ; V3 center; };
    std::vector<Cascade> cs(cascadeCount);
    // ---- per cascade up to the bounding sphere (the shader then needs WaveActiveMax over the cascades, :259)
    for (uint32_t cascadeId = 0; cascadeId < cascadeCount; cascadeId++) {
        float minZ, maxZ, splitLambda, splitStart;
        uint32_t splitCascadeCount, splitCascadeId;
        if (cascadeId < realtimeCount) {                                                                  // :108-141
            minZ = nearZ + cfg->cascadeStartDistance; maxZ = nearZ + cfg->cascadeEndDistance;
            splitLambda = cfg->splitLambda; splitCascadeCount = realtimeCount; splitCascadeId = cascadeId;
            if (validDepthMinMax) {
                float minZValid, maxZValid;
                std::memcpy(&minZValid, &validDepthMinMax[0], 4); std::memcpy(&maxZValid, &validDepthMinMax[1], 4);
                if (maxZValid > 0.0f) minZ = std::fmax(minZ, nearZ / maxZValid);
                if (minZValid > 0.0f) {
                    const float stableDistance = cfg->cascadeEndDistance - cfg->cascadeStartDistance;
                    maxZ = std::fmax(maxZ, minZ * 1.1f);
                    maxZ = std::fmin(maxZ, minZ + stableDistance);
                    maxZ = std::fmin(maxZ, nearZ / minZValid);
                }
            }
            splitStart = minZ - nearZ;
        } else {                                                                                          // :142-153
            maxZ = nearZ + cfg->farCascadeEndDistance; splitLambda = cfg->farCascadeSplitLambda;
            minZ = nearZ + cfg->cascadeEndDistance; splitStart = cfg->cascadeEndDistance;
            splitCascadeCount = cascadeCount - realtimeCount; splitCascadeId = cascadeId - realtimeCount;
        }
        const float splitDist = logCascadeSplit(maxZ, minZ, splitCascadeId, splitCascadeCount, splitLambda);
        const float prevSplitDist = splitCascadeId == 0 ? splitStart / clipRange
                                                        : logCascadeSplit(maxZ, minZ, splitCascadeId - 1, splitCascadeCount, splitLambda);
        const float splitDist_0 = logCascadeSplit(nearZ, nearZ + cfg->farCascadeEndDistance, 0, cascadeCount, cfg->farCascadeSplitLambda);   // :163 (argument order as written there)
        const float prevSplitDist_0 = 0.0f;
        V3 corner[8], corner0[8];
        static const float kNdc[8][3] = {{-1, 1, 1}, {1, 1, 1}, {1, -1, 1}, {-1, -1, 1}, {-1, 1, 0}, {1, 1, 0}, {1, -1, 0}, {-1, -1, 0}};
        for (int i = 0; i < 8; i++) {
            float h[4];
            mulPoint(view->clipToTranslatedWorldWithZFar_NoJitter, kNdc[i][0], kNdc[i][1], kNdc[i][2], 1.0f, h);
            corner[i] = {h[0] / h[3], h[1] / h[3], h[2] / h[3]};
        }
        for (int i = 0; i < 4; i++) {                                                                     // :181-191
            const V3 ray = sub(corner[i + 4], corner[i]);
            corner0[i + 4] = add(corner[i], mul(ray, splitDist_0));
            corner0[i] = add(corner[i], mul(ray, prevSplitDist_0));
        }
        for (int i = 0; i < 4; i++) {                                                                     // :194-203
            const V3 ray = sub(corner[i + 4], corner[i]);
            const V3 nearRay = mul(ray, prevSplitDist), farRay = mul(ray, splitDist);
            corner[i + 4] = add(corner[i], farRay);
            corner[i] = add(corner[i], nearRay);
        }
        V3 center = {0, 0, 0}, center0 = {0, 0, 0};
        for (int i = 0; i < 8; i++) { center = add(center, corner[i]); center0 = add(center0, corner0[i]); }
        center = {center.x / 8.0f, center.y / 8.0f, center.z / 8.0f};
        center0 = {center0.x / 8.0f, center0.y / 8.0f, center0.z / 8.0f};
        float sphereRadius = 0.0f, sphereRadius0 = 0.0f;
        for (int i = 0; i < 8; i++) {
            const V3 d = sub(corner[i], center), d0 = sub(corner0[i], center0);
            sphereRadius = std::fmax(sphereRadius, std::sqrt(dot(d, d)));
            sphereRadius0 = std::fmax(sphereRadius0, std::sqrt(dot(d0, d0)));
        }
        cs[cascadeId] = {sphereRadius, std::ceil(sphereRadius * 16.0f) / 16.0f, sphereRadius0, minZ, center};
    }
    float maxCascadeSphereRadius = 0.0f;                                                                  // WaveActiveMax, :259
    for (const Cascade& c : cs) maxCascadeSphereRadius = std::fmax(maxCascadeSphereRadius, c.cascadeSphereRadius);
    // ---- view, projection, texel snapping, planes
    for (uint32_t cascadeId = 0; cascadeId < cascadeCount; cascadeId++) {
        const Cascade& c = cs[cascadeId];
        const float maxE = c.cascadeSphereRadius, minE = -maxE;
        const float extentZ = maxCascadeSphereRadius * 2.0f;                                               // :267
        float radiusScale, zStartBiasScale;
        if (cascadeId >= realtimeCount) { radiusScale = c.sphereRadius0 / c.sphereRadius; zStartBiasScale = 1.0f; }
        else {
            radiusScale = 10.0f * cfg->radiusScaleFixed / c.sphereRadius;
            radiusScale = radiusScale / (radiusScale + 1.0f);
            const float startDistanceFactor = (c.minZ - nearZ) / (cfg->cascadeEndDistance - cfg->cascadeStartDistance);
            zStartBiasScale = 0.25f + startDistanceFactor;
        }
        radiusScale = std::fmin(radiusScale, 1.0f);
        const V3 lightDir = normalize(V3{lightDirIn[0], lightDirIn[1], lightDirIn[2]});
        const V3 shadowCameraPos = sub(c.center, mul(mul(lightDir, extentZ), 0.5f));                       // :294
        const float nearZProj = 0.0f, farZProj = extentZ;
        // lookAt_RH(eye, center, up), base.hlsli:637-666
        const V3 f = normalize(sub(c.center, shadowCameraPos));
        const V3 sv = normalize(cross(f, upDir));
        const V3 u = cross(sv, f);
        ChordMat4 shadowView;
        std::memset(&shadowView, 0, sizeof(shadowView));
        at(shadowView, 0, 0) = sv.x; at(shadowView, 0, 1) = sv.y; at(shadowView, 0, 2) = sv.z; at(shadowView, 0, 3) = -dot(sv, shadowCameraPos);
        at(shadowView, 1, 0) = u.x;  at(shadowView, 1, 1) = u.y;  at(shadowView, 1, 2) = u.z;  at(shadowView, 1, 3) = -dot(u, shadowCameraPos);
        at(shadowView, 2, 0) = -f.x; at(shadowView, 2, 1) = -f.y; at(shadowView, 2, 2) = -f.z; at(shadowView, 2, 3) = dot(f, shadowCameraPos);
        at(shadowView, 3, 3) = 1.0f;
        // ortho_RH_ZeroOne(left, right, bottom, top, zNear := farZProj, zFar := nearZProj), base.hlsli:668-682 (reverse Z)
        ChordMat4 shadowProj;
        std::memset(&shadowProj, 0, sizeof(shadowProj));
        {
            const float left = minE, right = maxE, bottom = minE, top = maxE, zn = farZProj, zf = nearZProj;
            at(shadowProj, 0, 0) = 2.0f / (right - left);
            at(shadowProj, 1, 1) = 2.0f / (top - bottom);
            at(shadowProj, 2, 2) = -1.0f / (zf - zn);
            at(shadowProj, 0, 3) = -(right + left) / (right - left);
            at(shadowProj, 1, 3) = -(top + bottom) / (top - bottom);
            at(shadowProj, 2, 3) = -zn / (zf - zn);
            at(shadowProj, 3, 3) = 1.0f;
        }
        // texel alignment, :312-326
        const float sMapSize = (float)cfg->cascadeDim;
        const ChordMat4 vp0 = matmul(shadowProj, shadowView);
        float origin[4];
        mulPoint(vp0, 0.0f, 0.0f, 0.0f, 1.0f, origin);
        for (int i = 0; i < 4; i++) origin[i] *= (sMapSize / 2.0f);
        const float roX = (std::nearbyint(origin[0]) - origin[0]) * (2.0f / sMapSize);                     // round(): half to even
        const float roY = (std::nearbyint(origin[1]) - origin[1]) * (2.0f / sMapSize);
        at(shadowProj, 0, 3) += roX;
        at(shadowProj, 1, 3) += roY;
        const ChordMat4 finalVP = matmul(shadowProj, shadowView);
        ChordMat4 reverseToWorld;
        if (!inverse4<float>(finalVP.m, reverseToWorld.m)) return CHORDVIS_E_INVALID;                      // (matrixInverse, base.hlsli:684-730: the same cofactor expansion)
        V3 p[8];
        for (int i = 0; i < 8; i++) {
            static const float kNdc[8][3] = {{-1, 1, 1}, {1, 1, 1}, {1, -1, 1}, {-1, -1, 1}, {-1, 1, 0}, {1, 1, 0}, {1, -1, 0}, {-1, -1, 0}};
            float h[4];
            mulPoint(reverseToWorld, kNdc[i][0], kNdc[i][1], kNdc[i][2], 1.0f, h);
            p[i] = {h[0] / h[3], h[1] / h[3], h[2] / h[3]};
        }
        float planes[6][4];
        auto plane = [&](int i, V3 a, V3 b, V3 o) {                                                        // normalize(cross(a - o, b - o)), -dot(n, o)
            const V3 n = normalize(cross(sub(a, o), sub(b, o)));
            planes[i][0] = n.x; planes[i][1] = n.y; planes[i][2] = n.z; planes[i][3] = -dot(n, o);
            return n;
        };
        plane(0, p[4], p[3], p[7]);                                                                        // left   :345-346
        plane(1, p[6], p[3], p[2]);                                                                        // down
        plane(2, p[6], p[1], p[5]);                                                                        // right
        plane(3, p[5], p[0], p[4]);                                                                        // top
        const V3 frontN = plane(4, p[1], p[3], p[0]);                                                      // front
        plane(5, p[5], p[7], p[6]);                                                                        // back
        planes[5][3] = -dot(frontN, p[6]);                                                                 // :365 uses frontN for the back plane's distance
        // isCascadeCacheValid, :8-22
        bool keep = false;
        if (bCacheValid && cascadeId >= realtimeCount) {
            const uint32_t period = cascadeCount - realtimeCount;
            keep = (tickCount % period) != (cascadeId - realtimeCount);
        }
        if (keep) continue;                                                                                // "Don't override view info."
        ChordInstanceCullingView& vi = views[cascadeId];
        std::memset(&vi, 0, sizeof(vi));
        vi.translatedWorldToClip = finalVP;
        vi.clipToTranslatedWorld = reverseToWorld;
        std::memcpy(vi.cameraWorldPos, mainInstanceView->cameraWorldPos, sizeof(vi.cameraWorldPos));
        vi.orthoDepthConvertToView[0] = at(shadowProj, 2, 2); vi.orthoDepthConvertToView[1] = at(shadowProj, 2, 3);
        vi.orthoDepthConvertToView[2] = zStartBiasScale; vi.orthoDepthConvertToView[3] = radiusScale;
        vi.renderDimension[0] = (float)cfg->cascadeDim; vi.renderDimension[1] = (float)cfg->cascadeDim;
        vi.renderDimension[2] = 1.0f / (float)cfg->cascadeDim; vi.renderDimension[3] = 1.0f / (float)cfg->cascadeDim;
        std::memcpy(vi.frustumPlanesRS, planes, sizeof(planes));
    }
    return CHORDVIS_OK;
}

} // extern "C"
