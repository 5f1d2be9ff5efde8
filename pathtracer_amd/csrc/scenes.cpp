// Built-in host scenes.
//
// PT_SCENE_MODEL restates the reference's active scene (src/main.cpp:800-840) with the model
// file as a parameter; PT_SCENE_RANDOM_SPHERES and PT_SCENE_CORNELL_BOX are the reference's two
// other scenes (src/main.cpp:516-625, 627-798).  The reference defines no multi-piece scene, so
// the chess scenes are this library's own: fixed, seedless constructions over the reference's
// six model files.
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>

#include "pt_internal.h"

using namespace pt;

namespace {

enum Piece { PAWN = 0, ROOK, KNIGHT, BISHOP, QUEEN, KING, PIECE_KINDS };
const char* const PIECE_FILES[PIECE_KINDS] = {"pawn", "rook", "knight", "bishop", "queen", "king"};

int load_model(const std::string& path, std::vector<PtTriangle>* out) {
    PtTriangle* data = nullptr;
    int64_t count = 0;
    const int rc = pt_load_triangles_file(path.c_str(), &data, &count);
    if (rc != PT_OK) return rc;
    out->assign(data, data + count);
    pt_free(data);
    if (out->empty()) return fail(PT_ERR_IO, "'%s' holds no triangles", path.c_str());
    return PT_OK;
}

// The model files are Z-up; the reference turns them Y-up with a rotation by -pi/2 about x
// built from float pi (src/main.cpp:808), so the result is not an exact axis swap.
PtMat3 model_upright() { return rotation(-PI_F / 2.0f, 1.0f, 0.0f, 0.0f); }

void append_transformed(const std::vector<PtTriangle>& model, const PtMat3& m, const PtVec3& offset, int material,
                        PtHostScene* scene) {
    for (const PtTriangle& t : model) {
        scene->triangles.push_back(PtTriangle{transform(m, t.a) + offset, transform(m, t.b) + offset, transform(m, t.c) + offset});
        scene->triangle_material_indices.push_back(material);
    }
}

int finish(PtHostScene* scene) {
    const int threads = host_threads();
    if (!scene->spheres.empty()) {
        std::vector<PtAabb> boxes;
        for (const PtSphere& s : scene->spheres) boxes.push_back(box_of(s));
        if (!sortable(boxes)) return fail(PT_ERR_INVALID_ARGUMENT, "a sphere has a NaN centre or radius: no BVH can be built");
        scene->sphere_bvh.resize(2 * boxes.size() - 1);
        build_bvh(boxes, scene->sphere_bvh.data(), threads);
    }
    if (!scene->triangles.empty()) {
        StageClock clock;
        std::vector<PtAabb> boxes(scene->triangles.size());
        in_parts(0, static_cast<long long>(boxes.size()), boxes.size() >= (1u << 16) ? host_threads() : 1, [&](long long from, long long to, int) {
            for (long long i = from; i < to; ++i) boxes[i] = box_of(scene->triangles[i]);
        });
        if (!sortable(boxes)) return fail(PT_ERR_INVALID_ARGUMENT, "a triangle has a NaN coordinate: no BVH can be built");
        clock.done("triangle boxes");
        scene->triangle_bvh.resize(2 * boxes.size() - 1);
        clock.done("node array allocated");
        build_bvh(boxes, scene->triangle_bvh.data(), threads);
        clock.done("reference-shaped BVH built");
    }
    return PT_OK;
}

// `focus` <= 0 focuses on the target (focus_distance = camera distance).
void set_camera(PtHostScene* scene, PtVec3 position, PtVec3 target, double fov_degrees, double aspect, double focus = 0.0) {
    pt_camera_look_at(&position, &target, degrees_to_radians(fov_degrees), 0.1f, focus, &scene->camera);
    scene->aspect_ratio = aspect;
}

// src/rng.cpp:63-69 and :30-32: the reference's xorshift32 step and its map to [0, 1]
struct SceneRng {
    uint32_t state;
    double next() {
        state ^= state << 13;
        state ^= state >> 17;
        state ^= state << 5;
        return static_cast<double>(state) / 4294967295.0;
    }
};

// 22 x 22 small spheres with hashed positions and materials on a 1000-unit ground sphere, plus
// three unit spheres (glass, brown, mirror); sphere k uses material k (src/main.cpp:516-625).
int build_random_spheres(PtHostScene* scene) {
    auto place = [scene](const PtMaterial& material, PtVec3 centre, double radius) {
        scene->sphere_material_indices.push_back(static_cast<int>(scene->materials.size()));
        scene->materials.push_back(material);
        scene->spheres.push_back(PtSphere{centre, radius});
    };
    place(lambertian(0.5f, 0.5f, 0.5f), PtVec3{0.0f, -1000.0f, 0.0f}, 1000.0f);         // :522-523
    SceneRng rng{479001599u};                                                            // :527
    for (int a = -11; a < 11; ++a) {
        for (int b = -11; b < 11; ++b) {
            const double material_choice = rng.next();                                   // :530-531
            const double x_offset = 0.9f * rng.next();                                   // :533-537
            const double z_offset = 0.9f * rng.next();
            const PtVec3 centre{static_cast<double>(a) + x_offset, 0.2f, static_cast<double>(b) + z_offset};
            if (material_choice < 0.8f) {                                                // :542-562
                const double r1 = rng.next(), g1 = rng.next(), b1 = rng.next();
                const double r2 = rng.next(), g2 = rng.next(), b2 = rng.next();
                place(lambertian(r1 * r2, g1 * g2, b1 * b2), centre, 0.2f);
            } else if (material_choice < 0.95f) {                                        // :563-576
                const double r = 0.5f * rng.next() + 0.5f;
                const double g = 0.5f * rng.next() + 0.5f;
                const double bl = 0.5f * rng.next() + 0.5f;
                const double fuzziness = rng.next();
                place(metal(r, g, bl, fuzziness), centre, 0.2f);
            } else {
                place(dielectric(1.5f), centre, 0.2f);                                   // :578
            }
        }
    }
    place(dielectric(1.5f), PtVec3{0.0f, 1.0f, 0.0f}, 1.0f);                             // :588-589
    place(lambertian(0.4f, 0.2f, 0.1f), PtVec3{-4.0f, 1.0f, 0.0f}, 1.0f);                // :593-594
    place(metal(0.7f, 0.6f, 0.5f, 0.0f), PtVec3{4.0f, 1.0f, 0.0f}, 1.0f);                // :598-599
    scene->background_start = PtColour{1.0f, 1.0f, 1.0f};                                // :614-615
    scene->background_end = PtColour{0.5f, 0.7f, 1.0f};
    set_camera(scene, PtVec3{13.0f, 2.0f, 3.0f}, PtVec3{0.0f, 0.0f, 0.0f}, 20.0f, 3.0f / 2.0f, 10.0f);   // :618-625, :33
    return finish(scene);
}

void append_quad(PtHostScene* scene, PtVec3 p0, PtVec3 p1, PtVec3 p2, PtVec3 p3, int material) {
    scene->triangles.push_back(PtTriangle{p0, p1, p2});
    scene->triangles.push_back(PtTriangle{p2, p3, p0});
    scene->triangle_material_indices.push_back(material);
    scene->triangle_material_indices.push_back(material);
}

// A unit cube scaled, turned about y and shifted, as 12 triangles in the reference's face
// order +z, +x, -z, -x, +y, -y (src/main.cpp:643-683).  Corner c has x = c&1, y = c>>1&1, z = c>>2.
void append_cornell_box(PtHostScene* scene, double height, double angle, PtVec3 shift, int material) {
    static const int FACES[6][4] = {{6, 4, 5, 7}, {7, 5, 1, 3}, {3, 1, 0, 2}, {2, 0, 4, 6}, {7, 3, 2, 6}, {4, 0, 1, 5}};
    const PtMat3 m = multiply(scaling(165.0f, height, 165.0f), rotation(angle, 0.0f, 1.0f, 0.0f));
    PtVec3 corner[8];
    for (int c = 0; c < 8; ++c) {
        corner[c] = transform(m, PtVec3{static_cast<double>(c & 1), static_cast<double>((c >> 1) & 1), static_cast<double>(c >> 2)}) + shift;
    }
    for (const int (&f)[4] : FACES) append_quad(scene, corner[f[0]], corner[f[1]], corner[f[2]], corner[f[3]], material);
}

// src/main.cpp:627-798
int build_cornell_box(PtHostScene* scene) {
    enum { RED = 0, WHITE = 1, GREEN = 2, LIGHT = 3, GLASS = 4 };
    scene->materials = {lambertian(0.65f, 0.05f, 0.05f), lambertian(0.73f, 0.73f, 0.73f), lambertian(0.12f, 0.45f, 0.15f),
                        diffuse_light(1.0f, 1.0f, 1.0f, 15.0f), dielectric(1.5f)};       // :627-633
    scene->spheres = {PtSphere{PtVec3{183.0f, 240.0f, 169.0f}, 75.0f}};                  // :635-639
    scene->sphere_material_indices = {GLASS};
    const double e = 555.0f;
    append_quad(scene, PtVec3{e, 0, 0}, PtVec3{e, 0, e}, PtVec3{e, e, e}, PtVec3{e, e, 0}, GREEN);    // :689-690
    append_quad(scene, PtVec3{0, 0, 0}, PtVec3{0, e, 0}, PtVec3{0, e, e}, PtVec3{0, 0, e}, RED);      // :693-694
    append_quad(scene, PtVec3{0, 0, 0}, PtVec3{0, 0, e}, PtVec3{e, 0, e}, PtVec3{e, 0, 0}, WHITE);    // :697-698 floor
    append_quad(scene, PtVec3{0, e, 0}, PtVec3{e, e, 0}, PtVec3{e, e, e}, PtVec3{0, e, e}, WHITE);    // :701-702 ceiling
    append_quad(scene, PtVec3{0, 0, e}, PtVec3{0, e, e}, PtVec3{e, e, e}, PtVec3{e, 0, e}, WHITE);    // :705-706 back
    append_quad(scene, PtVec3{213.0f, 554.0f, 227.0f}, PtVec3{343.0f, 554.0f, 227.0f}, PtVec3{343.0f, 554.0f, 332.0f},
                PtVec3{213.0f, 554.0f, 332.0f}, LIGHT);                                                // :709-710
    append_cornell_box(scene, 165.0f, -PI_F / 10.0f, PtVec3{130.0f, 0.0f, 65.0f}, WHITE);             // :669-673
    append_cornell_box(scene, 330.0f, PI_F / 12.0f, PtVec3{265.0f, 0.0f, 295.0f}, WHITE);             // :679-683
    scene->background_start = scene->background_end = PtColour{0.0f, 0.0f, 0.0f};       // :787-788
    set_camera(scene, PtVec3{278.0f, 278.0f, -800.0f}, PtVec3{278.0f, 278.0f, 0.0f}, 40.0f, 1.0f);    // :791-798, :34
    return finish(scene);
}

// src/main.cpp:800-840
int build_model(const char* path, PtHostScene* scene) {
    std::vector<PtTriangle> model;
    const int rc = load_model(path, &model);
    if (rc != PT_OK) return rc;
    scene->materials = {lambertian(6.0f / 255.0f, 4.0f / 255.0f, 3.0f / 255.0f),   // :801 float divisions, then widened
                        diffuse_light(1.0f, 1.0f, 1.0f, 10.0f)};                    // :802
    const PtMat3 upright = model_upright();
    for (const PtTriangle& t : model) {                                             // :809-813
        scene->triangles.push_back(PtTriangle{transform(upright, t.a), transform(upright, t.b), transform(upright, t.c)});
    }
    scene->triangle_material_indices.assign(model.size(), 0);                       // :806
    scene->spheres = {PtSphere{PtVec3{20.0f, 80.0f, 10.0f}, 20.0f}};                // :817
    scene->sphere_material_indices = {1};                                           // :818
    scene->background_start = scene->background_end = PtColour{0.01f, 0.01f, 0.01f};   // :829-830
    set_camera(scene, PtVec3{0.0f, 150.0f, 150.0f}, PtVec3{0.0f, 0.0f, 0.0f}, 40.0f, 1.0f);   // :833-840, :35
    return finish(scene);
}

int load_pieces(const char* dir, std::vector<PtTriangle> (&models)[PIECE_KINDS]) {
    for (int kind = 0; kind < PIECE_KINDS; ++kind) {
        const int rc = load_model(std::string(dir) + "/" + PIECE_FILES[kind] + ".triangles", &models[kind]);
        if (rc != PT_OK) return rc;
    }
    return PT_OK;
}

// One of each piece on a line: piece i (pawn, rook, knight, bishop, queen, king) stands at
// x = -100 + 40 i.  Materials, light and background are the reference model scene's.
int build_chess6_row(const char* dir, PtHostScene* scene) {
    std::vector<PtTriangle> models[PIECE_KINDS];
    const int rc = load_pieces(dir, models);
    if (rc != PT_OK) return rc;
    scene->materials = {lambertian(6.0f / 255.0f, 4.0f / 255.0f, 3.0f / 255.0f), diffuse_light(1.0f, 1.0f, 1.0f, 10.0f)};
    const PtMat3 upright = model_upright();
    for (int kind = 0; kind < PIECE_KINDS; ++kind) {
        append_transformed(models[kind], upright, PtVec3{-100.0f + 40.0f * kind, 0.0f, 0.0f}, 0, scene);
    }
    scene->spheres = {PtSphere{PtVec3{20.0f, 80.0f, 10.0f}, 20.0f}};
    scene->sphere_material_indices = {1};
    scene->background_start = scene->background_end = PtColour{0.01f, 0.01f, 0.01f};
    set_camera(scene, PtVec3{0.0f, 120.0f, 260.0f}, PtVec3{0.0f, 25.0f, 0.0f}, 40.0f, 640.0 / 360.0);
    return finish(scene);
}

// Material slots of the chess boards.
enum { M_BLACK = 0, M_LIGHT = 1, M_WHITE = 2, M_BOARD = 3, M_FIRST_MIXED = 4 };

// The 32-piece starting position on 40-unit squares centred on the origin: white on the
// +z side (towards the camera), black on the -z side and turned to face white.  The board
// is one quad (two triangles, normal +y) at the lowest vertex of any piece.
// `material_of(colour, kind)` chooses the material slot of a piece.
template <class MaterialOf>
void append_chess_set(const std::vector<PtTriangle> (&models)[PIECE_KINDS], const PtVec3& origin, MaterialOf material_of,
                      PtHostScene* scene) {
    static const Piece BACK_RANK[8] = {ROOK, KNIGHT, BISHOP, QUEEN, KING, BISHOP, KNIGHT, ROOK};
    const PtMat3 white = model_upright();
    const PtMat3 black = multiply(rotation(PI_F, 0.0f, 1.0f, 0.0f), white);
    double floor_y = 0.0;
    const size_t first = scene->triangles.size();
    for (int file = 0; file < 8; ++file) {
        const double x = -140.0f + 40.0f * file;
        append_transformed(models[BACK_RANK[file]], white, origin + PtVec3{x, 0.0f, 140.0f}, material_of(0, BACK_RANK[file]), scene);
        append_transformed(models[PAWN], white, origin + PtVec3{x, 0.0f, 100.0f}, material_of(0, PAWN), scene);
        append_transformed(models[PAWN], black, origin + PtVec3{x, 0.0f, -100.0f}, material_of(1, PAWN), scene);
        append_transformed(models[BACK_RANK[file]], black, origin + PtVec3{x, 0.0f, -140.0f}, material_of(1, BACK_RANK[file]), scene);
    }
    for (size_t i = first; i < scene->triangles.size(); ++i) {
        const PtTriangle& t = scene->triangles[i];
        floor_y = std::min(floor_y, std::min(t.a.y, std::min(t.b.y, t.c.y)));
    }
    const double half = 170.0f, y = floor_y;
    const PtVec3 p00 = origin + PtVec3{-half, y, -half}, p01 = origin + PtVec3{-half, y, half};
    const PtVec3 p11 = origin + PtVec3{half, y, half}, p10 = origin + PtVec3{half, y, -half};
    scene->triangles.push_back(PtTriangle{p00, p01, p11});
    scene->triangles.push_back(PtTriangle{p11, p10, p00});
    scene->triangle_material_indices.push_back(M_BOARD);
    scene->triangle_material_indices.push_back(M_BOARD);
}

void chess_base_materials(PtHostScene* scene) {
    scene->materials = {lambertian(6.0f / 255.0f, 4.0f / 255.0f, 3.0f / 255.0f),   // black pieces: the reference's albedo
                        diffuse_light(1.0f, 1.0f, 1.0f, 10.0f),
                        lambertian(0.73f, 0.73f, 0.73f),                           // white pieces
                        lambertian(0.4f, 0.3f, 0.2f)};                             // board
}

int build_chess32(const char* dir, bool mixed, int grid_n, PtHostScene* scene) {
    std::vector<PtTriangle> models[PIECE_KINDS];
    StageClock clock;
    const int rc = load_pieces(dir, models);
    if (rc != PT_OK) return rc;
    clock.done("piece models read");
    chess_base_materials(scene);
    if (mixed) {
        // every material type on some piece: kind -> slot M_FIRST_MIXED + 2*kind + colour
        const PtMaterial by_kind[PIECE_KINDS][2] = {
            {lambertian(0.8f, 0.75f, 0.6f), lambertian(0.15f, 0.1f, 0.1f)},          // pawn
            {metal(0.9f, 0.9f, 0.9f, 0.0f), metal(0.3f, 0.3f, 0.35f, 0.0f)},         // rook: mirror
            {metal(0.8f, 0.6f, 0.4f, 0.3f), metal(0.3f, 0.2f, 0.2f, 0.3f)},          // knight: brushed
            {dielectric(1.5f), dielectric(1.5f)},                                    // bishop: glass
            {lambertian(0.9f, 0.85f, 0.8f), lambertian(0.3f, 0.1f, 0.3f)},           // queen
            {metal(0.9f, 0.75f, 0.3f, 0.1f), metal(0.6f, 0.45f, 0.2f, 0.1f)}};       // king: gold
        for (int kind = 0; kind < PIECE_KINDS; ++kind) {
            scene->materials.push_back(by_kind[kind][0]);
            scene->materials.push_back(by_kind[kind][1]);
        }
        scene->materials[M_BOARD] = lambertian(0.5f, 0.5f, 0.5f);
        scene->background_start = PtColour{1.0f, 1.0f, 1.0f};                        // the sky of src/main.cpp:614-615
        scene->background_end = PtColour{0.5f, 0.7f, 1.0f};
    } else {
        scene->background_start = scene->background_end = PtColour{0.01f, 0.01f, 0.01f};
    }
    auto material_of = [mixed](int colour, int kind) {
        if (mixed) return M_FIRST_MIXED + 2 * kind + colour;
        return colour == 0 ? static_cast<int>(M_WHITE) : static_cast<int>(M_BLACK);
    };
    // One set per board of the lattice, in row order; the sets are made by several threads, each into arrays of its own,
    // and then put behind one another.
    const double pitch = 360.0f;
    const int sets = grid_n * grid_n;
    std::vector<PtHostScene> made(sets);
    in_parts(0, sets, std::min(sets, host_threads()), [&](long long from, long long to, int) {
        for (long long set = from; set < to; ++set) {
            const int gz = static_cast<int>(set) / grid_n, gx = static_cast<int>(set) % grid_n;
            const PtVec3 origin{pitch * (gx - 0.5 * (grid_n - 1)), 0.0f, pitch * (gz - 0.5 * (grid_n - 1))};
            append_chess_set(models, origin, material_of, &made[set]);
        }
    });
    std::vector<size_t> first(sets + 1, 0);
    for (int set = 0; set < sets; ++set) first[set + 1] = first[set] + made[set].triangles.size();
    scene->triangles.resize(first[sets]);
    scene->triangle_material_indices.resize(first[sets]);
    in_parts(0, sets, std::min(sets, host_threads()), [&](long long from, long long to, int) {
        for (long long set = from; set < to; ++set) {
            std::copy(made[set].triangles.begin(), made[set].triangles.end(), scene->triangles.begin() + first[set]);
            std::copy(made[set].triangle_material_indices.begin(), made[set].triangle_material_indices.end(), scene->triangle_material_indices.begin() + first[set]);
        }
    });
    made.clear();
    clock.done("chess sets placed");
    scene->spheres = {PtSphere{PtVec3{120.0f * grid_n, 330.0f * grid_n, 60.0f * grid_n}, 140.0f * grid_n}};
    scene->sphere_material_indices = {M_LIGHT};
    set_camera(scene, PtVec3{0.0f, 180.0f * grid_n, 330.0f * grid_n}, PtVec3{0.0f, 10.0f, 0.0f}, 40.0f, 16.0 / 9.0);
    return finish(scene);
}

}  // namespace

extern "C" {

int pt_host_scene_build(int32_t kind, const char* path, int32_t param, PtHostScene** out) {
    if (out == nullptr) return fail(PT_ERR_INVALID_ARGUMENT, "pt_host_scene_build: null output");
    *out = nullptr;
    std::unique_ptr<PtHostScene> scene(new PtHostScene());
    int rc;
    switch (kind) {
    case PT_SCENE_MODEL:
        if (path == nullptr) return fail(PT_ERR_INVALID_ARGUMENT, "PT_SCENE_MODEL needs a .triangles path");
        rc = build_model(path, scene.get());
        break;
    case PT_SCENE_CHESS6_ROW:
        if (path == nullptr) return fail(PT_ERR_INVALID_ARGUMENT, "chess scenes need the model directory");
        rc = build_chess6_row(path, scene.get());
        break;
    case PT_SCENE_CHESS32:
    case PT_SCENE_CHESS32_MIXED:
        if (path == nullptr) return fail(PT_ERR_INVALID_ARGUMENT, "chess scenes need the model directory");
        rc = build_chess32(path, kind == PT_SCENE_CHESS32_MIXED, 1, scene.get());
        break;
    case PT_SCENE_CHESS_GRID:
        if (path == nullptr) return fail(PT_ERR_INVALID_ARGUMENT, "chess scenes need the model directory");
        if (param < 1 || param > 8) return fail(PT_ERR_INVALID_ARGUMENT, "PT_SCENE_CHESS_GRID: grid_n must be in 1..8, got %d", param);
        rc = build_chess32(path, false, param, scene.get());
        break;
    case PT_SCENE_RANDOM_SPHERES:
        rc = build_random_spheres(scene.get());
        break;
    case PT_SCENE_CORNELL_BOX:
        rc = build_cornell_box(scene.get());
        break;
    default:
        return fail(PT_ERR_UNSUPPORTED, "pt_host_scene_build: unknown scene kind %d", kind);
    }
    if (rc != PT_OK) return rc;
    *out = scene.release();
    return PT_OK;
}

int pt_host_scene_desc(const PtHostScene* s, PtSceneDesc* d) {
    if (s == nullptr || d == nullptr) return fail(PT_ERR_INVALID_ARGUMENT, "pt_host_scene_desc: null argument");
    std::memset(d, 0, sizeof(*d));
    d->materials = s->materials.data();
    d->material_count = static_cast<int32_t>(s->materials.size());
    if (!s->spheres.empty()) {
        d->spheres = s->spheres.data();
        d->sphere_count = static_cast<int32_t>(s->spheres.size());
        d->sphere_bvh = s->sphere_bvh.data();
        d->sphere_bvh_node_count = static_cast<int32_t>(s->sphere_bvh.size());
        d->sphere_material_indices = s->sphere_material_indices.data();
    }
    if (!s->triangles.empty()) {
        d->triangles = s->triangles.data();
        d->triangle_count = static_cast<int32_t>(s->triangles.size());
        d->triangle_bvh = s->triangle_bvh.data();
        d->triangle_bvh_node_count = static_cast<int32_t>(s->triangle_bvh.size());
        d->triangle_material_indices = s->triangle_material_indices.data();
    }
    d->background_gradient_start = s->background_start;
    d->background_gradient_end = s->background_end;
    return PT_OK;
}

int pt_host_scene_camera(const PtHostScene* s, PtCamera* camera, double* aspect_ratio) {
    if (s == nullptr) return fail(PT_ERR_INVALID_ARGUMENT, "pt_host_scene_camera: null scene");
    if (camera != nullptr) *camera = s->camera;
    if (aspect_ratio != nullptr) *aspect_ratio = s->aspect_ratio;
    return PT_OK;
}

void pt_host_scene_destroy(PtHostScene* s) { delete s; }

}  // extern "C"
