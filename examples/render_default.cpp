// The reference's program (program/src/main.rs:18-149) over the C ABI of librt_hip.so, headless:
// the start-up scene of main.rs:21-66 assembled by the library's SceneBuilder, the event loop's
// render-and-advance turn (main.rs:129-149) run `frames` times, the image written as a PNG
// instead of being copied to a swapchain.  Uses nothing but include/rt_hip.h.
//
//   render_default <resources dir> <out.png> [frames=64] [width=1280] [height=720] [acc.bin] [--gpus N]
//
// With a sixth argument the raw accumulator (binding 7, RGBA32F) is written there as well.
// With --gpus N the event loop owns a group of N contexts instead of one (rt_group_*): the canvas is
// dealt to them in strips and assembled on the first GPU, the image is the same bit for bit.  A box
// with fewer than N GPUs names its GPUs more than once.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "../include/rt_hip.h"

namespace {

std::string slurp(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) {
        std::fprintf(stderr, "cannot read %s\n", path.c_str());
        std::exit(2);
    }
    return std::string(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
}

void check(rt_ctx* ctx, int rc, const char* what) {
    if (rc == RT_OK) return;
    std::fprintf(stderr, "%s failed (%d): %s\n", what, rc, rt_last_error(ctx));
    std::exit(1);
}

rt_affine3a srt(float sx, float sy, float sz, int axis, float angle, float tx, float ty, float tz) {
    float scale[3] = {sx, sy, sz}, quat[4] = {0.0f, 0.0f, 0.0f, 1.0f}, trans[3] = {tx, ty, tz};
    if (axis >= 0) rt_quat_from_rotation(axis, angle, quat);
    rt_affine3a out;
    rt_affine_from_srt(scale, quat, trans, &out);
    return out;
}

}  // namespace

int main(int argc, char** argv) {
    unsigned gpus = 1;
    for (int i = 1; i + 1 < argc; ++i)
        if (std::strcmp(argv[i], "--gpus") == 0) {
            gpus = (unsigned)std::atoi(argv[i + 1]);
            for (int k = i; k + 2 < argc; ++k) argv[k] = argv[k + 2];
            argc -= 2;
            break;
        }
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s <resources dir> <out.png> [frames] [width] [height] [acc.bin]\n", argv[0]);
        return 2;
    }
    const std::string res = argv[1];
    const unsigned frames = argc > 3 ? (unsigned)std::atoi(argv[3]) : 64u;
    const unsigned width = argc > 4 ? (unsigned)std::atoi(argv[4]) : 1280u;
    const unsigned height = argc > 5 ? (unsigned)std::atoi(argv[5]) : 720u;
    const float pi = 3.14159265358979323846f;  // f32::consts::PI

    // main.rs:21-34
    rt_cam_data cam;
    std::memset(&cam, 0, sizeof cam);
    cam.transform = srt(1.0f, 1.0f, 1.0f, -1, 0.0f, 0.0f, 0.0f, -20.0f);
    cam.canvas_width = width;
    cam.canvas_height = height;
    cam.fov = 90.0f;
    cam.depth = 10;
    cam.debug_number = 128;
    cam.debug_information = RT_DEBUG_NONE;

    // main.rs:36-66
    const rt_affine3a teapot = srt(0.6f * 2.0f, 1.01f * 2.0f, 0.6f * 2.0f, 0, pi, 0.0f, -9.8f, 0.0f);
    const rt_affine3a cube = srt(5.0f, 5.0f, 5.0f, 1, pi / 4.0f, 0.0f, 1.9f, 0.0f);
    const rt_affine3a box = srt(10.0f, 10.0f, 10.0f, -1, 0.0f, 0.0f, 0.0f, 0.0f);
    rt_scene_builder* sb = rt_scene_builder_new();
    rt_bvh_params params;
    rt_bvh_params_default(&params);
    const struct {
        const char* file;
        const rt_affine3a* xf;
    } meshes[3] = {{"default_cube.obj", &cube}, {"cornel_box.obj", &box}, {"teapot.obj", &teapot}};
    for (const auto& m : meshes) {
        const std::string text = slurp(res + "/" + m.file);
        if (rt_scene_builder_add_obj(sb, text.data(), text.size(), m.xf, 1, &params) != RT_OK) {
            std::fprintf(stderr, "unusable OBJ file %s\n", m.file);
            return 1;
        }
    }
    rt_scene_builder_sun_orientation(sb, 1.0f, -1.0f, 1.0f);
    rt_scene_info info;
    size_t nv = 0, nt = 0, nb = 0;
    if (rt_scene_builder_build(sb, &info, nullptr, &nv, nullptr, &nt, nullptr, nullptr, nullptr, nullptr, nullptr, &nb) != RT_OK)
        return 1;
    std::printf("merged: %zu vertices, %zu triangles, %zu BVH nodes\n", nv, nt, nb);  // main.rs:68-73

    rt_ctx* ctx = nullptr;
    rt_group* group = nullptr;
    auto gcheck = [&](int rc, const char* what) {
        if (rc == RT_OK) return;
        std::fprintf(stderr, "%s failed (%d): %s\n", what, rc, rt_group_last_error(group));
        std::exit(1);
    };
    if (gpus > 1) {
        int n_dev = 0;
        if (rt_device_count(&n_dev) != RT_OK || n_dev < 1) {
            std::fprintf(stderr, "no GPU\n");
            return 1;
        }
        std::vector<int> devices(gpus);
        for (unsigned i = 0; i < gpus; ++i) devices[i] = (int)(i % (unsigned)n_dev);
        gcheck(rt_group_create(devices.data(), devices.size(), &group), "rt_group_create");
        gcheck(rt_group_upload_scene_from_builder(group, sb), "rt_group_upload_scene_from_builder");
        ctx = rt_group_context(group, 0);  // holds the assembled image
        std::printf("group of %zu contexts on %d GPU(s)\n", rt_group_size(group), n_dev);
    } else {
        check(nullptr, rt_create(0, &ctx), "rt_create");
        check(ctx, rt_upload_scene_from_builder(ctx, sb), "rt_upload_scene_from_builder");
    }
    rt_scene_builder_free(sb);

    // the event loop's turn, main.rs:129-149: render, then frame += 1, frames_without_move += 1.
    // What the window would show is fetched the way a display loop wants it: rt_present_begin after
    // a render, rt_present_end one turn later, so that the copy of frame k runs beside the render
    // of frame k + 1.
    const auto t0 = std::chrono::steady_clock::now();
    const uint8_t* shown = nullptr;
    size_t shown_pixels = 0;
    unsigned long long checksum = 0;
    for (unsigned f = 0; f < frames; ++f) {
        if (group) gcheck(rt_group_render_frames(group, &cam, 1, RT_GROUP_TILES), "rt_group_render_frames");
        else check(ctx, rt_render_frames(ctx, &cam, 1), "rt_render_frames");
        if (f > 0) {
            check(ctx, rt_present_end(ctx, &shown, &shown_pixels), "rt_present_end");  // frame f - 1, "blitted"
            checksum += shown[0] + shown[4 * (shown_pixels / 2)];
        }
        check(ctx, rt_present_begin(ctx), "rt_present_begin");
    }
    if (frames > 0) check(ctx, rt_present_end(ctx, &shown, &shown_pixels), "rt_present_end");
    check(ctx, rt_synchronize(ctx), "rt_synchronize");
    const double total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    std::printf("%u frames of %ux%u: %.3f ms per displayed frame (%.1f fps), image bytes on the host every frame\n", frames,
                width, height, total_ms / (double)frames, 1000.0 * (double)frames / total_ms);
    (void)checksum;

    std::vector<float> image((size_t)width * height * 4);
    check(ctx, rt_read_present(ctx, image.data(), (size_t)width * height), "rt_read_present");
    if (rt_save_png(argv[2], image.data(), width, height) != RT_OK) {
        std::fprintf(stderr, "cannot write %s\n", argv[2]);
        return 1;
    }
    if (argc > 6) {
        check(ctx, rt_read_acc(ctx, image.data(), (size_t)width * height), "rt_read_acc");
        std::ofstream out(argv[6], std::ios::binary);
        out.write(reinterpret_cast<const char*>(image.data()), (std::streamsize)(image.size() * sizeof(float)));
    }
    if (group) rt_group_destroy(group);
    else rt_destroy(ctx);
    return 0;
}
