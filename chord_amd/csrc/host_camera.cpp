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

} // extern "C"
