// Generates tests/golden/glm_camera.json: the reference's camera math evaluated with the reference's
// own glm (external/include/glm, header-only, GLM 1.0.0) IN THIS CONTAINER ONLY.  The formulas are the
// ones at source/utils/camera.cpp:17-154, utils.cpp:186-198, viewport.cpp:267-268,434-445 and
// scene_node.cpp:42-90; this file is a fixture generator, not product code, and nothing from the
// reference travels with the fixture except numbers.
//
// build+run: see make_glm_fixture.sh
#include <cstdio>
#define GLM_FORCE_DEPTH_ZERO_TO_ONE
#include <glm/glm.hpp>
#include <glm/gtc/matrix_transform.hpp>

static void pm(const char* name, const glm::mat4& m, bool last = false)
{
    std::printf("    \"%s\": [", name);
    const float* p = &m[0][0];
    for (int i = 0; i < 16; i++) std::printf("%.9g%s", p[i], i < 15 ? ", " : "");
    std::printf("]%s\n", last ? "" : ",");
}

int main()
{
    struct Cam { double pos[3], front[3]; float fovy; double zn, zf; unsigned w, h; float jx, jy; };
    const Cam cams[] = {
        {{0, 0, 3}, {0, 0, -1}, glm::radians(45.0f), 0.001, 20000.0, 256, 256, 0.f, 0.f},
        {{-62, 12, 3}, {1.0, -0.18, -0.04}, glm::radians(45.0f), 0.001, 20000.0, 3840, 2160, 0.f, 0.f},
        {{25, 25, 25}, {-0.5, -0.6, 0.6}, glm::radians(60.0f), 0.01, 5000.0, 1920, 1080, 0.25f, -0.375f},
    };
    std::printf("[\n");
    for (unsigned ci = 0; ci < sizeof(cams) / sizeof(cams[0]); ci++) {
        const Cam& c = cams[ci];
        const glm::dvec3 front(c.front[0], c.front[1], c.front[2]), worldUp(0, 1, 0);
        const glm::dvec3 right = glm::normalize(glm::cross(front, worldUp));
        const glm::dvec3 up = glm::normalize(glm::cross(right, front));
        const glm::mat4 view = glm::lookAt(glm::vec3(0.0f), glm::vec3(front), glm::vec3(up));
        const float aspect = (float)c.w / (float)c.h;
        glm::mat4 proj(0.0f);
        const float t = tan(c.fovy * 0.5f);
        proj[0][0] = 1.0f / (aspect * t); proj[1][1] = 1.0f / t; proj[2][3] = -1.0f; proj[3][2] = (float)c.zn;
        glm::mat4 jit(1.0f);
        jit[3][0] += 2.0f * c.jx / (float)c.w;
        jit[3][1] += -2.0f * c.jy / (float)c.h;
        const glm::mat4 vp = (jit * proj) * view;
        // an object: rotate + non-uniform scale + translate (double), camera-relative then float
        glm::dmat4 l2w(1.0);
        l2w = glm::translate(l2w, glm::dvec3(12.5, -3.0, 40.25));
        l2w = glm::rotate(l2w, 0.7, glm::normalize(glm::dvec3(0.3, 1.0, -0.2)));
        l2w = glm::scale(l2w, glm::dvec3(1.5, 0.75, 2.0));
        glm::dmat4 rel = l2w;
        rel[3][0] -= c.pos[0]; rel[3][1] -= c.pos[1]; rel[3][2] -= c.pos[2];
        const glm::mat4 l2tw = glm::mat4(rel);
        const glm::mat4 tw2l = glm::inverse(l2tw);
        std::printf("  {\n    \"position\": [%.17g, %.17g, %.17g], \"front\": [%.17g, %.17g, %.17g],\n", c.pos[0], c.pos[1], c.pos[2], c.front[0], c.front[1], c.front[2]);
        std::printf("    \"fovy\": %.9g, \"zNear\": %.17g, \"zFar\": %.17g, \"width\": %u, \"height\": %u, \"jitter\": [%.9g, %.9g],\n", c.fovy, c.zn, c.zf, c.w, c.h, c.jx, c.jy);
        std::printf("    \"localToWorld\": [");
        for (int i = 0; i < 16; i++) std::printf("%.17g%s", (&l2w[0][0])[i], i < 15 ? ", " : "");
        std::printf("],\n");
        pm("translatedWorldToView", view);
        pm("translatedWorldToClip", vp);
        pm("clipToTranslatedWorld", glm::inverse(vp));
        pm("localToTranslatedWorld", l2tw);
        pm("translatedWorldToLocal", tw2l, true);
        std::printf("  }%s\n", ci + 1 < sizeof(cams) / sizeof(cams[0]) ? "," : "");
    }
    std::printf("]\n");
    return 0;
}
