#include "host_scene.hpp"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>

#include "host_math.hpp"

namespace rt {

// ---------------------------------------------------------------------------
// OBJ parsing, program/src/modules/mod.rs:6-36
// ---------------------------------------------------------------------------
namespace {

struct Tokens {
    std::vector<std::pair<const char*, size_t>> t;
};

void split_whitespace(const char* b, const char* e, Tokens& out) {
    out.t.clear();
    const char* p = b;
    while (p < e) {
        while (p < e && (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\f' || *p == '\v')) ++p;
        if (p >= e) break;
        const char* s = p;
        while (p < e && !(*p == ' ' || *p == '\t' || *p == '\r' || *p == '\f' || *p == '\v')) ++p;
        out.t.emplace_back(s, size_t(p - s));
    }
}

float parse_f32(const char* s, size_t n) {
    std::string tmp(s, n);
    return std::strtof(tmp.c_str(), nullptr);
}

// The index before the first '/', as the reference's parse::<i32> reads it.  Where the reference
// panics (not a number, out of range) this yields an index that no mesh has, which the checks on
// the finished mesh then refuse; the arithmetic on it is done in 64 bits so that nothing overflows.
int64_t parse_face_index(const char* s, size_t n) {
    size_t k = 0;
    while (k < n && s[k] != '/') ++k;
    std::string tmp(s, k);
    long long v = std::strtoll(tmp.c_str(), nullptr, 10);
    if (v > INT32_MAX) v = INT32_MAX;
    if (v < INT32_MIN) v = INT32_MIN;
    return v;
}

}  // namespace

void parse_obj(const char* text, size_t len, std::vector<rt_vertex>& vertices, std::vector<rt_tri>& faces) {
    const char* p = text;
    const char* end = text + len;
    Tokens tok;
    while (p < end) {
        const char* nl = static_cast<const char*>(std::memchr(p, '\n', size_t(end - p)));
        const char* le = nl ? nl : end;
        split_whitespace(p, le, tok);
        p = nl ? nl + 1 : end;
        if (tok.t.empty()) continue;
        const auto& head = tok.t[0];
        if (head.second == 1 && head.first[0] == 'v' && tok.t.size() >= 4) {
            rt_vertex v;
            std::memset(&v, 0, sizeof v);
            v.pos[0] = parse_f32(tok.t[1].first, tok.t[1].second);
            v.pos[1] = parse_f32(tok.t[2].first, tok.t[2].second);
            v.pos[2] = parse_f32(tok.t[3].first, tok.t[3].second);
            vertices.push_back(v);
        } else if (head.second == 1 && head.first[0] == 'f' && tok.t.size() >= 3) {
            // Fan triangulation of the polygon (mod.rs:19-31).  A negative
            // index is resolved against the number of tokens on the face
            // line, as the reference does (its inner `vertices` shadows the
            // vertex list, mod.rs:20, 25-27).
            const int64_t n_face = int64_t(tok.t.size() - 1);
            int64_t v1 = parse_face_index(tok.t[1].first, tok.t[1].second) - 1;
            int64_t prev = parse_face_index(tok.t[2].first, tok.t[2].second) - 1;
            for (size_t k = 3; k < tok.t.size(); ++k) {
                int64_t v2 = parse_face_index(tok.t[k].first, tok.t[k].second) - 1;
                rt_tri f;
                f.i0 = v1 < 0 ? uint32_t(n_face + v1) : uint32_t(v1);
                f.i1 = prev < 0 ? uint32_t(n_face + prev) : uint32_t(prev);
                f.i2 = v2 < 0 ? uint32_t(n_face + v2) : uint32_t(v2);
                faces.push_back(f);
                prev = v2;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// BVH construction, program/src/modules/bvh.rs
//
// Same decisions as the reference: candidate split positions
// round((i+1)*len/(splits+1)) for i < splits = len/4 on each axis of the
// centroid-sorted triangles (bvh.rs:102-116), cost = area*count of both sides
// (bvh.rs:109), first strictly smaller cost wins, children stored adjacently
// in depth-first creation order (bvh.rs:53-94).  Two things differ in how the
// same answer is reached: candidate boxes come from prefix / suffix sweeps
// instead of a fresh pass per candidate (min/max are exact, so the boxes are
// identical), and equal centroids keep their current order (a stable sort)
// where the reference's sort_unstable_by leaves the order unspecified.
// ---------------------------------------------------------------------------
namespace {

struct Box {
    V3 mn, mx;
};

inline Box empty_box() {
    return Box{v3(FLT_MAX, FLT_MAX, FLT_MAX), v3(-FLT_MAX, -FLT_MAX, -FLT_MAX)};  // bvh.rs:144-145
}
inline Box merge(const Box& a, const Box& b) { return Box{vmin(a.mn, b.mn), vmax(a.mx, b.mx)}; }

// bvh.rs:122-127
inline float surface_area(const Box& b) {
    float sx = std::fabs(b.mx.x - b.mn.x);
    float sy = std::fabs(b.mx.y - b.mn.y);
    float sz = std::fabs(b.mx.z - b.mn.z);
    return 2.0f * (sx * sy + sx * sz + sy * sz);
}

struct Item {
    rt_tri tri;
    float key[3];  // centroid per axis, bvh.rs:131-138
    Box box;       // bvh.rs:148-152
};

struct TNode {
    Box box;
    uint32_t first, last;  // triangle range, inclusive, LOCAL to the mesh
    TNode* c1 = nullptr;
    TNode* c2 = nullptr;
};

struct Arena {
    std::vector<std::unique_ptr<TNode[]>> blocks;
    size_t used = 0, cap = 0;
    TNode* alloc() {
        if (used == cap) {
            cap = 4096;
            used = 0;
            blocks.emplace_back(new TNode[cap]);
        }
        return &blocks.back()[used++];
    }
};

struct BuildCtx {
    BvhBuildParams params;
    std::atomic<int> spare_threads{0};
};

void sort_by_axis(Item* items, size_t n, int axis) {
    std::stable_sort(items, items + n, [axis](const Item& a, const Item& b) { return a.key[axis] < b.key[axis]; });
}

Box bounding_box(const Item* items, size_t n) {
    Box b = empty_box();
    for (size_t i = 0; i < n; ++i) b = merge(b, items[i].box);
    return b;
}

// bvh.rs:97-120
void find_ideal_split(Item* items, size_t n, size_t splits, int& best_axis, size_t& best_split,
                      std::vector<Box>& suffix) {
    float best_result = FLT_MAX;
    best_axis = 0;
    best_split = 0;
    suffix.resize(n + 1);
    for (int axis = 0; axis < 3; ++axis) {
        sort_by_axis(items, n, axis);
        suffix[n] = empty_box();
        for (size_t i = n; i-- > 0;) suffix[i] = merge(items[i].box, suffix[i + 1]);
        // suffix[] was folded right-to-left; min/max are exact and
        // associative, so the box equals the left-to-right fold of bvh.rs:147-155.
        float chunk_size = float(n) / float(splits + 1);
        Box prefix = empty_box();
        size_t covered = 0;
        for (size_t i = 0; i < splits; ++i) {
            size_t split_index = size_t(std::round((float(i) + 1.0f) * chunk_size));
            if (split_index > n) split_index = n;
            while (covered < split_index) prefix = merge(prefix, items[covered++].box);
            float split_cost =
                surface_area(prefix) * float(split_index) + surface_area(suffix[split_index]) * (float(n) - float(split_index));
            if (split_cost < best_result) {
                best_result = split_cost;
                best_axis = axis;
                best_split = split_index;
            }
        }
    }
}

void build_recursive(BuildCtx& ctx, Arena& arena, Item* items, size_t n, uint32_t start, TNode* parent,
                     uint32_t depth);

void build_child(BuildCtx& ctx, Arena& arena, Item* items, size_t n, uint32_t start, TNode* node, uint32_t depth) {
    if (n > ctx.params.leaf_max && depth < ctx.params.max_depth) build_recursive(ctx, arena, items, n, start, node, depth + 1);
}

// bvh.rs:30-95
void build_recursive(BuildCtx& ctx, Arena& arena, Item* items, size_t n, uint32_t start, TNode* parent,
                     uint32_t depth) {
    int axis;
    size_t split;
    {
        std::vector<Box> scratch;
        find_ideal_split(items, n, n / 4, axis, split, scratch);
    }
    sort_by_axis(items, n, axis);
    if (split == 0 || split >= n) return;  // no usable candidate: the parent stays a leaf

    TNode* c1 = arena.alloc();
    TNode* c2 = arena.alloc();
    c1->box = bounding_box(items, split);
    c1->first = start;
    c1->last = start + uint32_t(split) - 1;
    c2->box = bounding_box(items + split, n - split);
    c2->first = start + uint32_t(split);
    c2->last = start + uint32_t(n) - 1;
    parent->c1 = c1;
    parent->c2 = c2;

    // The two sides touch disjoint triangle ranges; large first halves go to a
    // helper thread (with its own arena) when one is free.
    bool spawned = false;
    std::thread helper;
    std::unique_ptr<Arena> helper_arena;
    if (split > 32768) {
        int avail = ctx.spare_threads.load();
        while (avail > 0 && !ctx.spare_threads.compare_exchange_weak(avail, avail - 1)) {
        }
        if (avail > 0) {
            spawned = true;
            helper_arena.reset(new Arena);
            Arena* ha = helper_arena.get();
            helper = std::thread(
                [&ctx, ha, items, split, start, c1, depth] { build_child(ctx, *ha, items, split, start, c1, depth); });
        }
    }
    if (!spawned) build_child(ctx, arena, items, split, start, c1, depth);
    build_child(ctx, arena, items + split, n - split, start + uint32_t(split), c2, depth);
    if (spawned) {
        helper.join();
        ctx.spare_threads.fetch_add(1);
        // keep the helper's nodes alive for the emit pass
        for (auto& b : helper_arena->blocks) arena.blocks.insert(arena.blocks.begin(), std::move(b));
    }
}

rt_bvh make_leaf(const TNode* n) {
    rt_bvh out;
    std::memset(&out, 0, sizeof out);
    st(out.bounding_box.min, n->box.mn);
    st(out.bounding_box.max, n->box.mx);
    out.child_1_or_first_tri = n->first;
    out.child_2_or_last_tri = n->last;
    out.mode = RT_MODE_TRIANGLES;
    return out;
}

// Numbers nodes the way the reference's recursion does: both children are
// appended when their parent is split, then the first subtree, then the second.
void emit_children(std::vector<rt_bvh>& out, const TNode* node, uint32_t index) {
    // explicit stack: (node, index) pairs whose children still have to be appended
    struct Todo {
        const TNode* n;
        uint32_t idx;
    };
    std::vector<Todo> todo;
    todo.push_back(Todo{node, index});
    while (!todo.empty()) {
        Todo t = todo.back();
        todo.pop_back();
        if (!t.n->c1) continue;
        uint32_t k = uint32_t(out.size());
        out.push_back(make_leaf(t.n->c1));
        out.push_back(make_leaf(t.n->c2));
        out[t.idx].child_1_or_first_tri = k;
        out[t.idx].child_2_or_last_tri = k + 1;
        out[t.idx].mode = RT_MODE_CHILDREN;
        todo.push_back(Todo{t.n->c2, k + 1});  // popped second
        todo.push_back(Todo{t.n->c1, k});      // popped first: its whole subtree precedes c2's
    }
}

}  // namespace

bool mesh_is_buildable(const rt_vertex* vertices, size_t n_vertices, const rt_tri* triangles, size_t n_triangles) {
    for (size_t i = 0; i < n_triangles; ++i)
        if (triangles[i].i0 >= n_vertices || triangles[i].i1 >= n_vertices || triangles[i].i2 >= n_vertices) return false;
    for (size_t i = 0; i < n_vertices; ++i)
        for (int k = 0; k < 3; ++k)
            if (!std::isfinite(vertices[i].pos[k])) return false;
    return true;
}

std::vector<rt_bvh> create_bvh(const rt_vertex* vertices, size_t n_vertices, rt_tri* triangles, size_t n_triangles,
                               const BvhBuildParams& params) {
    (void)n_vertices;
    std::vector<rt_bvh> out;
    if (n_triangles == 0) return out;
    std::vector<Item> items(n_triangles);
    for (size_t i = 0; i < n_triangles; ++i) {
        const rt_tri& t = triangles[i];
        V3 a = ld(vertices[t.i0].pos), b = ld(vertices[t.i1].pos), c = ld(vertices[t.i2].pos);
        items[i].tri = t;
        items[i].key[0] = ((a.x + b.x) + c.x) / 3.0f;
        items[i].key[1] = ((a.y + b.y) + c.y) / 3.0f;
        items[i].key[2] = ((a.z + b.z) + c.z) / 3.0f;
        items[i].box = Box{vmin(vmin(a, b), c), vmax(vmax(a, b), c)};
    }
    Arena arena;
    TNode* root = arena.alloc();
    root->box = bounding_box(items.data(), n_triangles);
    root->first = 0;
    root->last = uint32_t(n_triangles - 1);
    BuildCtx ctx;
    ctx.params = params;
    ctx.spare_threads.store(params.threads > 1 ? params.threads - 1 : 0);
    if (n_triangles > params.root_leaf_max) build_recursive(ctx, arena, items.data(), n_triangles, 0, root, 1);

    out.reserve(n_triangles / 4 + 16);
    out.push_back(make_leaf(root));
    emit_children(out, root, 0);
    for (size_t i = 0; i < n_triangles; ++i) triangles[i] = items[i].tri;
    return out;
}

uint32_t bvh_depth(const rt_bvh* nodes, size_t n_nodes, uint32_t root, uint32_t limit) {
    if (root >= n_nodes) return 0;
    struct E {
        uint32_t node, depth;
    };
    std::vector<E> stack;
    stack.push_back(E{root, 1});
    uint32_t best = 0;
    size_t visited = 0;
    while (!stack.empty()) {
        E e = stack.back();
        stack.pop_back();
        if (++visited > n_nodes) return 0;  // more visits than nodes: not a tree
        if (e.depth > limit) return 0;
        best = std::max(best, e.depth);
        const rt_bvh& n = nodes[e.node];
        if (n.mode == RT_MODE_CHILDREN) {
            if (n.child_1_or_first_tri >= n_nodes || n.child_2_or_last_tri >= n_nodes) return 0;
            stack.push_back(E{n.child_1_or_first_tri, e.depth + 1});
            stack.push_back(E{n.child_2_or_last_tri, e.depth + 1});
        }
    }
    return best;
}

// ---------------------------------------------------------------------------
// SceneBuilder, program/src/modules/mod.rs:38-134
// ---------------------------------------------------------------------------
SceneBuilder::SceneBuilder() {
    sun[0] = 1.0f;
    sun[1] = -1.0f;
    sun[2] = 1.0f;  // mod.rs:56
}

void SceneBuilder::add_mesh(std::vector<rt_vertex> mesh_vertices, std::vector<rt_tri> mesh_tris,
                            const rt_affine3a* transforms, size_t n_transforms, const BvhBuildParams& params) {
    std::vector<rt_bvh> mesh_bvh =
        create_bvh(mesh_vertices.data(), mesh_vertices.size(), mesh_tris.data(), mesh_tris.size(), params);
    add_built_mesh(mesh_vertices, mesh_tris, mesh_bvh, transforms, n_transforms);
}

// mod.rs:61-104: the mesh's indices move behind what the scene already holds
void SceneBuilder::add_built_mesh(const std::vector<rt_vertex>& mesh_vertices, const std::vector<rt_tri>& mesh_tris,
                                  const std::vector<rt_bvh>& mesh_bvh, const rt_affine3a* transforms,
                                  size_t n_transforms) {
    const uint32_t vert_offset = uint32_t(vertices.size());
    const uint32_t bvh_offset = uint32_t(bvh.size());
    const uint32_t tri_offset = uint32_t(tris.size());
    const uint32_t object_offset = uint32_t(objects.size());

    vertices.insert(vertices.end(), mesh_vertices.begin(), mesh_vertices.end());
    tris.reserve(tris.size() + mesh_tris.size());
    for (const rt_tri& t : mesh_tris) tris.push_back(rt_tri{t.i0 + vert_offset, t.i1 + vert_offset, t.i2 + vert_offset});
    bvh.reserve(bvh.size() + mesh_bvh.size());
    for (rt_bvh node : mesh_bvh) {
        if (node.mode == RT_MODE_CHILDREN) {
            node.child_1_or_first_tri += bvh_offset;
            node.child_2_or_last_tri += bvh_offset;
        } else {
            node.child_1_or_first_tri += tri_offset;
            node.child_2_or_last_tri += tri_offset;
        }
        bvh.push_back(node);
    }
    objects.push_back(rt_object{bvh_offset});
    for (size_t i = 0; i < n_transforms; ++i) {
        rt_instance inst;
        std::memset(&inst, 0, sizeof inst);
        inst.transform = transforms[i];
        inst.object_id = object_offset;
        instances.push_back(inst);
    }
}

void SceneBuilder::add_obj_file(const char* text, size_t len, const rt_affine3a* transforms, size_t n_transforms,
                                const BvhBuildParams& params) {
    std::vector<rt_vertex> v;
    std::vector<rt_tri> f;
    parse_obj(text, len, v, f);
    add_mesh(std::move(v), std::move(f), transforms, n_transforms, params);
}

void SceneBuilder::sun_orientation(float x, float y, float z) {
    sun[0] = x;
    sun[1] = y;
    sun[2] = z;
}

rt_scene_info SceneBuilder::info() const {
    rt_scene_info s;
    s.num_instances = uint32_t(instances.size());
    s.num_bvh_nodes = uint32_t(bvh.size());
    s.num_triangles = uint32_t(tris.size());
    s.sun_orientation[0] = sun[0];
    s.sun_orientation[1] = sun[1];
    s.sun_orientation[2] = sun[2];
    return s;
}

}  // namespace rt
