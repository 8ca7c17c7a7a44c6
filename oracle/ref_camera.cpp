/*
 * ref_camera.cpp — host harness around the REFERENCE'S OWN interactive camera
 * (common/misc.hpp:108-224 CameraControl) and camera set-up (common/camera.hpp:11-25).
 *
 * TEST INFRASTRUCTURE ONLY. No restated algorithm: the reference's headers are #included where they
 * lie under $(REF) and compiled as the reference compiles them for the host (plain C++, its own
 * float3 of common/math.hpp:7-21, glibc sinf/cosf/tan). Built by oracle/Makefile into
 * oracle/_ref/ref_camera. GLFW/GL appear as headers only (declarations of the window callbacks'
 * parameter types); nothing of them is called.
 *
 * stdin:  eye.xyz lookat.xyz W H fovy                      (floats as decimal, ints)
 *         then any number of events "button dx dy"        (button 0 = left/orbit, 1 = right/zoom, 2 = middle/pan)
 * stdout: per event one line of hex float bits: eye.xyz lookat.xyz updated  raygen origin.xyz right.xyz up.xyz
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "common/misc.hpp"
#include "common/camera.hpp"

static unsigned bits(float f)
{
    unsigned u;
    memcpy(&u, &f, 4);
    return u;
}

int main()
{
    float e[3], a[3], fovy;
    int W, H;
    if (scanf("%f %f %f %f %f %f %d %d %f", &e[0], &e[1], &e[2], &a[0], &a[1], &a[2], &W, &H, &fovy) != 9) return 2;
    CameraControl cc;
    cc.m_cameraOrig = {e[0], e[1], e[2]};
    cc.m_cameraLookat = {a[0], a[1], a[2]};
    int button;
    float dx, dy;
    while (scanf("%d %f %f", &button, &dx, &dy) == 3)
    {
        /* one drag event of (dx, dy) pixels with one button held: previous cursor position 0 */
        cc.m_isInit = true;
        cc.m_xpos = 0.0f;
        cc.m_ypos = 0.0f;
        cc.mouseButtonCallback(nullptr, button, GLFW_PRESS, 0);
        cc.cursorPosCallback(nullptr, (double)dx, (double)dy);
        cc.mouseButtonCallback(nullptr, button, GLFW_RELEASE, 0);
        const bool upd = cc.is_updated();
        const float3 o = cc.cameraOrigin(), l = cc.cameraLookAt();
        RayGenerator rg;
        rg.lookat(o, l, {0.0f, 1.0f, 0.0f}, fovy, W, H); /* 10_restir_di.cpp:249-251 */
        printf("%08x %08x %08x %08x %08x %08x %d %08x %08x %08x %08x %08x %08x %08x %08x %08x\n", bits(o.x), bits(o.y), bits(o.z), bits(l.x),
               bits(l.y), bits(l.z), upd ? 1 : 0, bits(rg.m_origin.x), bits(rg.m_origin.y), bits(rg.m_origin.z), bits(rg.m_right.x),
               bits(rg.m_right.y), bits(rg.m_right.z), bits(rg.m_up.x), bits(rg.m_up.y), bits(rg.m_up.z));
    }
    return 0;
}
