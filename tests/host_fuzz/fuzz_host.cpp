// Host-side fuzz driver, meant to be built with AddressSanitizer and UBSan (tests/host_fuzz/Makefile):
// mutated OBJ text through rt_parse_obj, random and degenerate meshes through rt_create_bvh, and both
// through the scene builder.  Checks only what must hold for any input: a return code, no out-of-range
// index in what comes back, no memory error.  Needs no GPU.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/rt_hip.h"

static uint64_t g_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() {
    g_state ^= g_state << 13;
    g_state ^= g_state >> 7;
    g_state ^= g_state << 17;
    return (uint32_t)(g_state >> 16);
}
static float rndf(float lo, float hi) { return lo + (hi - lo) * (float)(rnd() & 0xFFFFFF) / 16777215.0f; }

static std::string slurp(const char* path) {
    std::ifstream f(path, std::ios::binary);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

#define CHECK(cond)                                                            \
    do {                                                                       \
        if (!(cond)) {                                                         \
            std::fprintf(stderr, "CHECK failed line %d: %s\n", __LINE__, #cond); \
            std::exit(1);                                                      \
        }                                                                      \
    } while (0)

static void check_tree(const std::vector<rt_bvh>& nodes, size_t n_nodes, size_t n_tris) {
    for (size_t i = 0; i < n_nodes; ++i) {
        const rt_bvh& b = nodes[i];
        if (b.mode == RT_MODE_TRIANGLES) {
            // an empty range (first = last + 1) is legal
            CHECK(b.child_1_or_first_tri <= n_tris && (b.child_2_or_last_tri < n_tris || b.child_2_or_last_tri + 1 == b.child_1_or_first_tri));
        } else {
            CHECK(b.child_1_or_first_tri < n_nodes && b.child_2_or_last_tri < n_nodes);
        }
    }
}

// parse (two-call pattern), and when the text parses, build a tree over it
static int parse_and_build(const std::string& text, uint32_t leaf_max, int threads) {
    size_t nv = 0, nt = 0;
    int rc = rt_parse_obj(text.data(), text.size(), nullptr, &nv, nullptr, &nt);
    if (rc != RT_OK) return rc;
    std::vector<rt_vertex> v(nv + 1);
    std::vector<rt_tri> t(nt + 1);
    size_t nv2 = nv, nt2 = nt;
    rc = rt_parse_obj(text.data(), text.size(), v.data(), &nv2, t.data(), &nt2);
    CHECK(rc == RT_OK && nv2 == nv && nt2 == nt);
    if (nt == 0 || nv == 0) return RT_OK;
    rt_bvh_params p;
    rt_bvh_params_default(&p);
    p.leaf_max = leaf_max;
    p.threads = threads;
    size_t n_nodes = 0;
    rc = rt_create_bvh(v.data(), nv, t.data(), nt, &p, nullptr, &n_nodes);
    if (rc != RT_OK) return rc;  // e.g. an index beyond the vertices
    std::vector<rt_bvh> nodes(n_nodes);
    size_t n2 = n_nodes;
    rc = rt_create_bvh(v.data(), nv, t.data(), nt, &p, nodes.data(), &n2);
    CHECK(rc == RT_OK && n2 == n_nodes);
    check_tree(nodes, n2, nt);
    return RT_OK;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? std::atoi(argv[1]) : 300;
    const char* dir = argc > 2 ? argv[2] : "ray_tracing_amd/resources";
    std::vector<std::string> seeds;
    for (const char* name : {"default_cube.obj", "cornel_box.obj", "teapot.obj"}) {
        std::string s = slurp((std::string(dir) + "/" + name).c_str());
        CHECK(!s.empty());
        if (s.size() > 20000) s.resize(20000);  // a prefix of the teapot is plenty
        seeds.push_back(s);
    }
    seeds.push_back("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nf 1 2 3 4\nf -1 -2 -3\nf 1/1/1 2//2 3/3\n");
    int parsed = 0, refused = 0;
    static const char* const junk[] = {"f", "v", "f 1", "f 1 2", "v 1 2", "v a b c", "f 0 0 0", "f 99999999999 1 2", "f -99 1 2",
                                       "v 1e39 -1e39 nan", "v inf 0 0", "f 1/ 2/ 3/", "f / / /", "\r", "\0", "f 1 2 3 4 5 6 7 8 9",
                                       "v 1 2 3 4 5", "#", "vt 1 2", "vn 0 0 1", "f 2147483648 1 2", "f -2147483648 1 2"};
    for (int r = 0; r < rounds; ++r) {
        std::string s = seeds[rnd() % seeds.size()];
        const int n_mut = 1 + (int)(rnd() % 6);
        for (int m = 0; m < n_mut; ++m) {
            const size_t at = s.empty() ? 0 : rnd() % s.size();
            switch (rnd() % 5) {
                case 0:
                    if (!s.empty()) s[at] = (char)(rnd() & 0xFF);
                    break;
                case 1:
                    s.insert(at, std::string("\n") + junk[rnd() % (sizeof junk / sizeof *junk)] + "\n");
                    break;
                case 2:
                    s.erase(at, rnd() % 40);
                    break;
                case 3:
                    s.resize(at);
                    break;
                default:
                    s.insert(at, s.substr(rnd() % (s.size() + 1), rnd() % 200));
            }
        }
        const int rc = parse_and_build(s, 1 + rnd() % 20, 1 + (int)(rnd() % 4));
        CHECK(rc == RT_OK || rc == RT_ERR_INVALID || rc == RT_ERR_LIMIT);
        (rc == RT_OK ? parsed : refused)++;
        // the same text through the scene builder, twice, with a couple of transforms
        rt_scene_builder* b = rt_scene_builder_new();
        rt_affine3a xf[2];
        const float one[3] = {1, 1, 1}, q[4] = {0, 0, 0, 1}, tr[3] = {rndf(-5, 5), 0, 0};
        rt_affine_from_srt(one, q, tr, &xf[0]);
        xf[1] = xf[0];
        rt_bvh_params p;
        rt_bvh_params_default(&p);
        for (int k = 0; k < 2; ++k) {
            const int rb = rt_scene_builder_add_obj(b, s.data(), s.size(), xf, 1 + k, &p);
            CHECK(rb == RT_OK || rb == RT_ERR_INVALID || rb == RT_ERR_LIMIT);
        }
        rt_scene_info info;
        const rt_vertex* pv;
        const rt_tri* pt;
        const rt_object* po;
        const rt_instance* pi;
        const rt_bvh* pb;
        size_t nv, nt, no, ni, nb;
        const int rb = rt_scene_builder_build(b, &info, &pv, &nv, &pt, &nt, &po, &no, &pi, &ni, &pb, &nb);
        if (rb == RT_OK) {
            for (size_t i = 0; i < nt; ++i) CHECK(pt[i].i0 < nv && pt[i].i1 < nv && pt[i].i2 < nv);
            for (size_t i = 0; i < ni; ++i) CHECK(pi[i].object_id < no);
            for (size_t i = 0; i < no; ++i) CHECK(po[i].bvh_root < nb);
        }
        rt_scene_builder_free(b);
    }
    // random meshes: soups, duplicates, degenerate triangles, coincident points, extreme coordinates
    for (int r = 0; r < rounds; ++r) {
        const size_t nv = 1 + rnd() % 200, nt = 1 + rnd() % 400;
        std::vector<rt_vertex> v(nv);
        const int flavour = (int)(rnd() % 5);
        for (auto& x : v) {
            std::memset(&x, 0, sizeof x);
            for (int k = 0; k < 3; ++k) {
                float c = rndf(-10, 10);
                if (flavour == 1) c = std::round(c);            // many coincident points
                if (flavour == 2) c *= 1.0e30f;                 // overflowing areas
                if (flavour == 3 && k == 1) c = 2.5f;           // flat
                if (flavour == 4) c = 0.0f;                     // one point
                x.pos[k] = c;
            }
        }
        std::vector<rt_tri> t(nt);
        for (auto& f : t) {
            f.i0 = rnd() % nv;
            f.i1 = rnd() % nv;
            f.i2 = (rnd() % 7 == 0) ? f.i0 : rnd() % nv;
        }
        rt_bvh_params p;
        rt_bvh_params_default(&p);
        p.leaf_max = 1 + rnd() % 24;
        p.root_leaf_max = rnd() % 8;
        p.max_depth = 1 + rnd() % 40;
        p.threads = 1 + (int)(rnd() % 4);
        size_t n_nodes = 0;
        int rc = rt_create_bvh(v.data(), nv, t.data(), nt, &p, nullptr, &n_nodes);
        CHECK(rc == RT_OK || rc == RT_ERR_INVALID || rc == RT_ERR_LIMIT);
        if (rc != RT_OK) continue;
        const std::vector<rt_tri> t0 = t;
        // too small a node buffer must be refused, not overrun, and must leave the triangles alone
        if (n_nodes > 1) {
            size_t small = n_nodes - 1;
            std::vector<rt_bvh> few(small);
            rc = rt_create_bvh(v.data(), nv, t.data(), nt, &p, few.data(), &small);
            CHECK(rc != RT_OK && std::memcmp(t.data(), t0.data(), nt * sizeof(rt_tri)) == 0);
        }
        std::vector<rt_bvh> nodes(n_nodes);
        size_t n2 = n_nodes;
        rc = rt_create_bvh(v.data(), nv, t.data(), nt, &p, nodes.data(), &n2);
        CHECK(rc == RT_OK && n2 == n_nodes);  // the counting call saw the same triangles
        check_tree(nodes, n2, nt);
    }
    // ready-made trees with arbitrary content through add_built_mesh: refused or taken, never trusted
    int taken = 0;
    for (int r = 0; r < rounds; ++r) {
        const size_t nv = 3 + rnd() % 30, nt = 1 + rnd() % 40, nn = 1 + rnd() % 30;
        std::vector<rt_vertex> v(nv);
        for (auto& x : v) {
            std::memset(&x, 0, sizeof x);
            for (int k = 0; k < 3; ++k) x.pos[k] = rndf(-4, 4);
        }
        std::vector<rt_tri> t(nt);
        for (auto& f : t) {
            f.i0 = rnd() % (nv + (rnd() % 9 == 0 ? 3 : 0));  // now and then out of range
            f.i1 = rnd() % nv;
            f.i2 = rnd() % nv;
        }
        std::vector<rt_bvh> nodes(nn);
        const bool sane = rnd() % 2 == 0;
        for (size_t i = 0; i < nn; ++i) {
            rt_bvh& b = nodes[i];
            std::memset(&b, 0, sizeof b);
            for (int k = 0; k < 3; ++k) {
                b.bounding_box.min[k] = rndf(-5, 0);
                b.bounding_box.max[k] = rndf(0, 5);
            }
            if (sane && 2 * i + 2 < nn) {  // a heap-shaped tree
                b.mode = RT_MODE_CHILDREN;
                b.child_1_or_first_tri = (uint32_t)(2 * i + 1);
                b.child_2_or_last_tri = (uint32_t)(2 * i + 2);
            } else if (sane) {
                b.mode = RT_MODE_TRIANGLES;
                b.child_1_or_first_tri = rnd() % nt;
                b.child_2_or_last_tri = b.child_1_or_first_tri + rnd() % (nt - b.child_1_or_first_tri);
            } else {
                b.mode = rnd() % 3;
                b.child_1_or_first_tri = rnd() % (2 * nn + 2);
                b.child_2_or_last_tri = rnd() % 5 == 0 ? rnd() : rnd() % (2 * nn + 2);
            }
        }
        rt_scene_builder* b = rt_scene_builder_new();
        rt_affine3a xf;
        const float one[3] = {1, 1, 1}, q[4] = {0, 0, 0, 1}, tr[3] = {0, 0, 0};
        rt_affine_from_srt(one, q, tr, &xf);
        const int rc = rt_scene_builder_add_built_mesh(b, v.data(), nv, t.data(), nt, nodes.data(), nn, &xf, 1);
        CHECK(rc == RT_OK || rc == RT_ERR_INVALID || rc == RT_ERR_LIMIT);
        if (rc == RT_OK) {
            ++taken;
            rt_scene_info info;
            const rt_vertex* pv;
            const rt_tri* pt;
            const rt_object* po;
            const rt_instance* pi;
            const rt_bvh* pb;
            size_t cv, ct, co, ci, cb;
            CHECK(rt_scene_builder_build(b, &info, &pv, &cv, &pt, &ct, &po, &co, &pi, &ci, &pb, &cb) == RT_OK);
            for (size_t i = 0; i < ct; ++i) CHECK(pt[i].i0 < cv && pt[i].i1 < cv && pt[i].i2 < cv);
            std::vector<rt_bvh> got(pb, pb + cb);
            check_tree(got, cb, ct);
        }
        rt_scene_builder_free(b);
    }
    // the image writers on canvases of every small size, with values no display can show
    {
        const std::string png = "/tmp/rt_host_fuzz.png", exr = "/tmp/rt_host_fuzz.exr";
        const int n_images = rounds < 60 ? rounds : 60;
        for (int r = 0; r < n_images; ++r) {
            const uint32_t w = 1 + rnd() % 70, h = 1 + rnd() % 50;
            std::vector<float> img((size_t)w * h * 4);
            for (auto& x : img) {
                switch (rnd() % 12) {
                    case 0: x = NAN; break;
                    case 1: x = INFINITY; break;
                    case 2: x = -INFINITY; break;
                    case 3: x = -rndf(0, 10); break;
                    case 4: x = rndf(1, 1.0e30f); break;
                    default: x = rndf(0, 1);
                }
            }
            CHECK(rt_save_png(png.c_str(), img.data(), w, h) == RT_OK);
            CHECK(rt_save_exr(exr.c_str(), img.data(), w, h, rndf(0, 2)) == RT_OK);
            CHECK(slurp(png.c_str()).size() > 8 && slurp(exr.c_str()).size() > (size_t)w * h * 12);
        }
        CHECK(rt_save_png("/nonexistent-directory/x.png", nullptr, 4, 4) != RT_OK);
        float one[4] = {0, 0, 0, 1};
        CHECK(rt_save_png("/nonexistent-directory/x.png", one, 1, 1) == RT_ERR_IO);
        CHECK(rt_save_png(png.c_str(), one, 0, 1) != RT_OK);
        std::remove(png.c_str());
        std::remove(exr.c_str());
    }
    // one mesh large enough for the host builder to hand subtrees to helper threads (the race
    // detector's part of the job): the tree must not depend on the number of threads
    if (argc > 3) {
        const size_t side = (size_t)std::atoi(argv[3]);
        std::vector<rt_vertex> v((side + 1) * (side + 1));
        for (size_t i = 0; i <= side; ++i)
            for (size_t j = 0; j <= side; ++j) {
                rt_vertex& x = v[i * (side + 1) + j];
                std::memset(&x, 0, sizeof x);
                x.pos[0] = (float)j;
                x.pos[1] = rndf(0, 3);
                x.pos[2] = (float)i;
            }
        std::vector<rt_tri> t0;
        for (size_t i = 0; i < side; ++i)
            for (size_t j = 0; j < side; ++j) {
                const uint32_t a = (uint32_t)(i * (side + 1) + j), b = a + 1, c = a + (uint32_t)side + 1, d = c + 1;
                t0.push_back(rt_tri{a, b, c});
                t0.push_back(rt_tri{b, d, c});
            }
        std::vector<rt_bvh> first;
        std::vector<rt_tri> first_t;
        for (int threads : {1, 4, 3}) {
            std::vector<rt_tri> t = t0;
            rt_bvh_params p;
            rt_bvh_params_default(&p);
            p.leaf_max = 4;
            p.threads = threads;
            size_t n_nodes = 2 * t.size();
            std::vector<rt_bvh> nodes(n_nodes);
            CHECK(rt_create_bvh(v.data(), v.size(), t.data(), t.size(), &p, nodes.data(), &n_nodes) == RT_OK);
            nodes.resize(n_nodes);
            check_tree(nodes, n_nodes, t.size());
            if (threads == 1) {
                first = nodes;
                first_t = t;
            } else {
                CHECK(nodes.size() == first.size() && std::memcmp(nodes.data(), first.data(), nodes.size() * sizeof(rt_bvh)) == 0);
                CHECK(std::memcmp(t.data(), first_t.data(), t.size() * sizeof(rt_tri)) == 0);
            }
        }
        std::printf("threaded build of %zu triangles: the same tree with 1, 4 and 3 threads\n", t0.size());
    }
    std::printf("ready-made trees: %d of %d taken\n", taken, rounds);
    std::printf("host fuzz ok: %d texts parsed, %d refused, %d random meshes\n", parsed, refused, rounds);
    return 0;
}
