// Lifecycle stress of a context through the C ABI with the host side of the library (context, layout,
// scene builder) under AddressSanitizer and UBSan; the kernels come unsanitized from the product
// build.  Runs on a GPU box: random sequences of uploads, option changes, renders on canvases of
// changing size, read-backs, the two-step present, instance and material updates, partitions, and
// destroy / create.  Only valid scenes are ever rendered.  Checks return codes and that what is read
// back has the expected size and a sane alpha; the sanitizers check the rest.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/rt_hip.h"

static uint64_t g_state = 0xA0761D6478BD642Full;
static uint32_t rnd() {
    g_state ^= g_state << 13;
    g_state ^= g_state >> 7;
    g_state ^= g_state << 17;
    return (uint32_t)(g_state >> 16);
}

#define CHECK(cond)                                                                                      \
    do {                                                                                                 \
        if (!(cond)) {                                                                                   \
            std::fprintf(stderr, "CHECK failed line %d: %s (%s)\n", __LINE__, #cond, rt_last_error(ctx)); \
            std::exit(1);                                                                                \
        }                                                                                                \
    } while (0)

static std::string slurp(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? std::atoi(argv[1]) : 200;
    const std::string dir = argc > 2 ? argv[2] : "ray_tracing_amd/resources";
    const std::string cube = slurp(dir + "/default_cube.obj"), box = slurp(dir + "/cornel_box.obj");
    rt_ctx* ctx = nullptr;
    CHECK(!cube.empty() && !box.empty());
    CHECK(rt_create(0, &ctx) == RT_OK);

    rt_cam_data cam;
    std::memset(&cam, 0, sizeof cam);
    size_t n_inst = 0;
    bool have_scene = false;
    uint32_t rendered_w = 0, rendered_h = 0;  // canvas of the last render, 0: nothing rendered since the last change
    bool begun = false;
    bool probe_on = false, counters_on = false, probed = false, counted = false;  // options, and whether a render has run with them
    std::vector<rt_instance> inst_copy;

    auto new_cam = [&](uint32_t w, uint32_t h) {
        std::memset(&cam, 0, sizeof cam);
        cam.transform.x_axis[0] = cam.transform.y_axis[1] = cam.transform.z_axis[2] = 1.0f;
        cam.transform.translation[2] = -20.0f;
        cam.canvas_width = w;
        cam.canvas_height = h;
        cam.fov = 90.0f;
        cam.depth = 1 + rnd() % 6;
    };
    auto upload = [&]() {
        rt_scene_builder* b = rt_scene_builder_new();
        const int n_meshes = 1 + (int)(rnd() % 2);
        n_inst = 0;
        for (int m = 0; m < n_meshes; ++m) {
            rt_affine3a xf[2];
            for (auto& x : xf) {
                const float s = m == 0 ? 5.0f : 10.0f;
                const float sc[3] = {s, s, s}, q[4] = {0, 0, 0, 1}, tr[3] = {(float)(rnd() % 5) - 2.0f, 0, 0};
                rt_affine_from_srt(sc, q, tr, &x);
            }
            const size_t k = 1 + rnd() % 2;
            const std::string& text = m == 0 ? cube : box;
            rt_bvh_params p;
            rt_bvh_params_default(&p);
            if (rnd() % 2) CHECK(rt_scene_builder_use_bvh_builder(b, rnd() % 2 ? RT_BVH_DEVICE_SAH : RT_BVH_DEVICE_LINEAR, 0) == RT_OK);
            else CHECK(rt_scene_builder_use_bvh_builder(b, RT_BVH_HOST, 0) == RT_OK);
            CHECK(rt_scene_builder_add_obj(b, text.data(), text.size(), xf, k, &p) == RT_OK);
            n_inst += k;
        }
        CHECK(rt_upload_scene_from_builder(ctx, b) == RT_OK);
        {
            rt_scene_info info;
            const rt_vertex* pv;
            const rt_tri* pt;
            const rt_object* po;
            const rt_instance* pi;
            const rt_bvh* pb;
            size_t cv, ct, co, ci, cb;
            CHECK(rt_scene_builder_build(b, &info, &pv, &cv, &pt, &ct, &po, &co, &pi, &ci, &pb, &cb) == RT_OK && ci == n_inst);
            inst_copy.assign(pi, pi + ci);
        }
        rt_scene_builder_free(b);
        have_scene = true;
    };

    upload();
    new_cam(64, 36);
    std::vector<float> img;
    std::vector<uint8_t> img8;
    for (int r = 0; r < rounds; ++r) {
        switch (rnd() % 17) {
            case 0:
                upload();
                break;
            case 1: {  // another canvas
                new_cam(8 + rnd() % 120, 8 + rnd() % 70);
                rendered_w = rendered_h = 0;
                begun = probed = counted = false;
                break;
            }
            case 2:
            case 3: {
                const uint32_t n = 1 + rnd() % 3;
                CHECK(rt_render_frames(ctx, &cam, n) == RT_OK);
                rendered_w = cam.canvas_width;
                rendered_h = cam.canvas_height;
                probed = probe_on;
                counted = counters_on;
                break;
            }
            case 4:
                CHECK(rt_render_frame(ctx, &cam) == RT_OK);
                rendered_w = cam.canvas_width;
                rendered_h = cam.canvas_height;
                probed = probe_on;
                counted = counters_on;
                break;
            case 5:
                if (rendered_w) {
                    img.assign((size_t)rendered_w * rendered_h * 4, -1.0f);
                    CHECK(rt_read_acc(ctx, img.data(), (size_t)rendered_w * rendered_h) == RT_OK);
                    CHECK(rt_read_present(ctx, img.data(), (size_t)rendered_w * rendered_h) == RT_OK);
                    CHECK(img[3] == 1.0f || img[3] == 0.0f);  // 0: a row of another rank's share, never rendered here
                    CHECK(rt_read_acc(ctx, img.data(), (size_t)rendered_w * rendered_h + 1) != RT_OK);  // wrong size
                }
                break;
            case 6:
                if (rendered_w) {
                    img8.assign((size_t)rendered_w * rendered_h * 4, 7);
                    CHECK(rt_read_present_rgba8(ctx, img8.data(), (size_t)rendered_w * rendered_h) == RT_OK);
                    CHECK(img8[3] == 255 || img8[3] == 0);
                }
                break;
            case 7:
                if (rendered_w) {
                    CHECK(rt_present_begin(ctx) == RT_OK);
                    begun = true;
                }
                break;
            case 8:
                if (begun) {
                    const uint8_t* p = nullptr;
                    size_t n = 0;
                    CHECK(rt_present_end(ctx, &p, &n) == RT_OK && p && n == (size_t)rendered_w * rendered_h);
                    CHECK((p[3] == 255 || p[3] == 0) && (p[4 * n - 1] == 255 || p[4 * n - 1] == 0));
                }
                break;
            case 9: {
                const int opt[] = {RT_OPT_PIPELINE, RT_OPT_PROBE_HITS, RT_OPT_COUNTERS, RT_OPT_PRESENT, RT_OPT_TILE_ORDER, RT_OPT_TILE_DEAL};
                const int o = opt[rnd() % 6];
                const int64_t v = o == RT_OPT_PIPELINE ? rnd() % 3 : o == RT_OPT_TILE_DEAL ? (int64_t)(rnd() % 3) - 1 : rnd() % 2;
                CHECK(rt_set_option(ctx, o, v) == RT_OK);
                if (o == RT_OPT_PROBE_HITS) probe_on = v != 0, probed = false;
                if (o == RT_OPT_COUNTERS) counters_on = v != 0, counted = false;
                if (o == RT_OPT_PRESENT && v == 0) {
                    rendered_w = rendered_h = 0;  // no presented image until the option is back on and a frame rendered
                    begun = false;
                    CHECK(rt_set_option(ctx, RT_OPT_PRESENT, 1) == RT_OK);
                }
                break;
            }
            case 10: {
                const uint32_t world = 1 + rnd() % 4;
                CHECK(rt_set_rows(ctx, 8, rnd() % world, world) == RT_OK);
                break;
            }
            case 11: {
                std::vector<rt_material> m(n_inst);
                for (size_t i = 0; i < n_inst; ++i) rt_default_material((uint32_t)(rnd() % 4), &m[i]);
                CHECK(rt_set_materials(ctx, m.data(), n_inst) == RT_OK);
                CHECK(rt_set_materials(ctx, m.data(), n_inst + 1) != RT_OK);
                if (rnd() % 2) CHECK(rt_set_materials(ctx, nullptr, 0) == RT_OK);
                break;
            }
            case 12: {
                rt_stats st;
                CHECK(rt_get_stats(ctx, &st) == RT_OK);
                CHECK(rt_reset_stats(ctx) == RT_OK);
                CHECK(rt_clear_acc(ctx) == RT_OK);
                CHECK(rt_synchronize(ctx) == RT_OK);
                break;
            }
            case 13: {  // the instances move (what the reference does by refilling its instance buffer)
                for (auto& x : inst_copy) x.transform.translation[1] = (float)(rnd() % 7) - 3.0f;
                CHECK(rt_update_instances(ctx, inst_copy.data(), inst_copy.size()) == RT_OK);
                CHECK(rt_update_instances(ctx, inst_copy.data(), inst_copy.size() + 1) != RT_OK);
                break;
            }
            case 14:
                if (rendered_w && probed && counted) {
                    const size_t n = (size_t)rendered_w * rendered_h;
                    std::vector<uint32_t> a(n), b2(n), c2(n);
                    CHECK(rt_read_hits(ctx, a.data(), b2.data(), c2.data(), n) == RT_OK);
                    CHECK(rt_read_counters(ctx, a.data(), b2.data(), n) == RT_OK);
                    CHECK(rt_read_hits(ctx, a.data(), b2.data(), c2.data(), n - 1) != RT_OK);
                }
                break;
            case 15: {  // the caller's own accumulator, on a canvas of its own
                const uint32_t w = 8 + rnd() % 40, h = 8 + rnd() % 30;
                std::vector<float> acc((size_t)w * h * 4, 0.25f);
                CHECK(rt_write_acc(ctx, acc.data(), (size_t)w * h, w, h) == RT_OK);
                CHECK(rt_write_acc(ctx, acc.data(), (size_t)w * h + 1, w, h) != RT_OK);
                new_cam(w, h);
                cam.frames_without_move = 1.0f;
                rendered_w = rendered_h = 0;
                begun = probed = counted = false;
                break;
            }
            default: {  // a new context now and then
                if (rnd() % 4 == 0) {
                    rt_destroy(ctx);
                    ctx = nullptr;
                    CHECK(rt_create(0, &ctx) == RT_OK);
                    have_scene = false;
                    rendered_w = rendered_h = 0;
                    begun = probed = counted = probe_on = counters_on = false;
                    CHECK(rt_render_frame(ctx, &cam) == RT_ERR_NO_SCENE);
                    upload();
                }
            }
        }
    }
    CHECK(have_scene);
    rt_destroy(ctx);
    std::printf("context fuzz ok: %d steps\n", rounds);
    std::fflush(stdout);
    return 0;
}
