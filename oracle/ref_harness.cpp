// TEST INFRASTRUCTURE — not part of the product.
//
// Builds the reference renderer's own hot-path sources (read in place from
// /root/reference/src at compile time, never copied) into oracle/_ref/libptref.so so
// that the CPU restatement in pt_oracle.cpp and the HIP path can be checked against
// the reference's real arithmetic.  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may load the result.
//
// What comes from the reference, unmodified (unity order of src/main.cpp:1-18):
//   linear_algebra, path_tracing, geometry, material, colour, bvh, rng (.h + .cpp).
// What cannot be compiled here (both need <Windows.h>) and is therefore restated
// below, each piece citing the lines it follows:
//   src/main.cpp           render_scanline body, camera frame math, model scene
//                          constants, tone-map
//   src/model_loading.cpp  load_triangles_file (CreateFileA/ReadFile -> fread)
//
// Build: see oracle/Makefile (g++ -std=c++17 -O2 -fwrapv -ffp-contract=off, asserts live
// exactly like build.bat:3, which never defines NDEBUG).

#include <algorithm>
#include <atomic>
#include <cassert>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

// ---- the reference, in its own unity order (src/main.cpp:1-18, minus model_loading) ----
#include "linear_algebra.h"
#include "path_tracing.h"
#include "geometry.h"
#include "material.h"
#include "colour.h"
#include "types.h"
#include "bvh.h"
#include "rng.h"

#include "linear_algebra.cpp"
#include "path_tracing.cpp"
#include "geometry.cpp"
#include "material.cpp"
#include "colour.cpp"
#include "bvh.cpp"
#include "rng.cpp"

#include "../include/pathtracer_amd.h"

static_assert(sizeof(Vec3) == sizeof(PtVec3), "Vec3");
static_assert(sizeof(Mat3) == sizeof(PtMat3), "Mat3");
static_assert(sizeof(Ray) == sizeof(PtRay), "Ray");
static_assert(sizeof(AABB) == sizeof(PtAabb), "AABB");
static_assert(sizeof(Sphere) == sizeof(PtSphere), "Sphere");
static_assert(sizeof(Triangle) == sizeof(PtTriangle), "Triangle");
static_assert(sizeof(Node) == sizeof(PtBvhNode) && sizeof(Node) == 64, "Node");
static_assert(sizeof(Material) == sizeof(PtMaterial) && sizeof(Material) == 40, "Material");
static_assert(offsetof(Material, type) == offsetof(PtMaterial, type), "Material.type");
static_assert(sizeof(Camera) == sizeof(PtCamera) && sizeof(Camera) == 128, "Camera");
static_assert(sizeof(Colour) == sizeof(PtColour), "Colour");
static_assert(offsetof(Node, index) == offsetof(PtBvhNode, index), "Node.index");

// src/main.cpp:27-30
static constexpr real PI = 3.14159265358979323846264f;
static real degrees_to_radians(const real degrees) {
    return degrees / 180.0f * PI;
}

struct RefScene {
    std::vector<Material> materials;
    std::vector<Sphere> spheres;
    std::vector<int> sphere_material_indices;
    BVH sphere_bvh;
    std::vector<Triangle> triangles;
    std::vector<int> triangle_material_indices;
    BVH triangle_bvh;
    Scene scene;
};

static void finish_scene(RefScene* s, const Colour& bg_start, const Colour& bg_end) {
    if (!s->spheres.empty()) {
        s->sphere_bvh = construct_sphere_bvh(s->spheres.data(), static_cast<int>(s->spheres.size()));
    }
    if (!s->triangles.empty()) {
        s->triangle_bvh = construct_triangle_bvh(s->triangles.data(), static_cast<int>(s->triangles.size()));
    }
    s->scene = Scene{
        s->materials.data(),
        s->spheres.empty() ? nullptr : s->spheres.data(),
        s->spheres.empty() ? nullptr : &s->sphere_bvh,
        s->spheres.empty() ? nullptr : s->sphere_material_indices.data(),
        s->triangles.empty() ? nullptr : s->triangles.data(),
        s->triangles.empty() ? nullptr : &s->triangle_bvh,
        s->triangles.empty() ? nullptr : s->triangle_material_indices.data(),
        bg_start,
        bg_end
    };
}

// The per-pixel body of render_scanline (src/main.cpp:88-121) with CLIENT_WIDTH /
// CLIENT_HEIGHT (src/main.cpp:38-39) as run-time values.
static Ray primary_ray(const PtFrame& f, const int row, const int column, const int sample) {
    const real aperture = f.aperture;
    const Vec3 camera_position{f.camera_position.x, f.camera_position.y, f.camera_position.z};
    const Vec3 camera_x{f.camera_x.x, f.camera_x.y, f.camera_x.z};
    const Vec3 camera_y{f.camera_y.x, f.camera_y.y, f.camera_y.z};
    const Vec3 bottom_left{f.bottom_left.x, f.bottom_left.y, f.bottom_left.z};
    const Vec3 step_x{f.step_x.x, f.step_x.y, f.step_x.z};
    const Vec3 step_y{f.step_y.x, f.step_y.y, f.step_y.z};

    const real lens_radius = 0.5f * aperture;                                       // :87
    u32 rng = noise_3d(row, column, sample);                                        // :89

    rng = random_number(rng);                                                       // :91
    const real u_randomness = real_from_rng(rng);
    const real u = (static_cast<real>(column) + u_randomness) / static_cast<real>(f.width - 1);   // :93

    rng = random_number(rng);
    const real v_randomness = real_from_rng(rng);
    const real v = (static_cast<real>(row) + v_randomness) / static_cast<real>(f.height - 1);     // :97

    rng = random_number(rng);
    const real a_randomness = real_from_rng(rng);
    const real a = aperture * a_randomness - lens_radius;                           // :101

    const real b_max = std::sqrt(lens_radius * lens_radius - a * a);                // :103
    const real b_min = -b_max;

    rng = random_number(rng);
    const real b_randomness = real_from_rng(rng);
    const real b = (b_max - b_min) * b_randomness + b_min;                          // :108

    const Vec3 random_offset = a * camera_x + b * camera_y;                         // :110

    const Vec3 ray_direction = normalise(Vec3{bottom_left + u * step_x + v * step_y - camera_position - random_offset});  // :112
    return Ray{camera_position + random_offset, ray_direction};                     // :113
}

static void render_row(const RefScene& s, const PtFrame& f, const int row, real* const pixels) {
    for (int sample = f.sample_begin; sample < f.sample_begin + f.sample_count; ++sample) {
        for (int column = 0; column < f.width; ++column) {
            const Ray ray = primary_ray(f, row, column, sample);
            const Colour colour = intersect(ray, s.scene);                          // :115
            const long long index = 4ll * (static_cast<long long>(row) * f.width + column);   // :117
            pixels[index + 0] += colour.b;
            pixels[index + 1] += colour.g;
            pixels[index + 2] += colour.r;
            pixels[index + 3] += 1.0f;
        }
    }
}

extern "C" {

// ---------------------------------------------------------------- known-answer hooks
unsigned ref_noise_1d(int x) { return noise_1d(x); }
unsigned ref_noise_3d(int x, int y, int z) { return noise_3d(x, y, z); }
unsigned ref_random_number(unsigned seed) { return random_number(seed); }
double ref_real_from_rng(unsigned rng) { return real_from_rng(rng); }

void ref_random_unit_vector(const double* seed, double* out) {
    const Vec3 r = random_unit_vector(Vec3{seed[0], seed[1], seed[2]});
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

static Ray ray_from(const double* p) { return Ray{Vec3{p[0], p[1], p[2]}, Vec3{p[3], p[4], p[5]}}; }

// NB: the reference asserts |direction|^2 ~ 1 in each test; callers pass unit directions.
void ref_ray_aabb(const double* ray, const double* aabb, double* out) {
    const AABBIntersections r = intersect(ray_from(ray), AABB{Vec3{aabb[0], aabb[1], aabb[2]}, Vec3{aabb[3], aabb[4], aabb[5]}});
    out[0] = r.min_distance; out[1] = r.max_distance;
}

void ref_ray_sphere(const double* ray, const double* sphere, double* out) {
    const Maybe<SphereIntersections> r = intersect(ray_from(ray), Sphere{Vec3{sphere[0], sphere[1], sphere[2]}, sphere[3]});
    out[0] = r.is_valid ? 1.0 : 0.0; out[1] = r.value.min_distance; out[2] = r.value.max_distance;
}

void ref_ray_triangle(const double* ray, const double* tri, double* out) {
    const Triangle t{Vec3{tri[0], tri[1], tri[2]}, Vec3{tri[3], tri[4], tri[5]}, Vec3{tri[6], tri[7], tri[8]}};
    const Maybe<real> r = intersect(ray_from(ray), t);
    out[0] = r.is_valid ? 1.0 : 0.0; out[1] = r.value;
}

void ref_unit_normal(const double* tri, double* out) {
    const Triangle t{Vec3{tri[0], tri[1], tri[2]}, Vec3{tri[3], tri[4], tri[5]}, Vec3{tri[6], tri[7], tri[8]}};
    const Vec3 n = unit_normal(t);
    out[0] = n.x; out[1] = n.y; out[2] = n.z;
}

// in: ray[6], material v[4] + type (as double), point[3], normal[3]; out: valid, ray[6]
void ref_scatter(const double* ray, const double* material, const double* point, const double* normal, double* out) {
    Material m = {};
    std::memcpy(&m, material, 32);
    m.type = static_cast<Material::Type>(static_cast<int>(material[4]));
    const Maybe<Ray> r = scatter(ray_from(ray), m, Vec3{point[0], point[1], point[2]}, Vec3{normal[0], normal[1], normal[2]});
    out[0] = r.is_valid ? 1.0 : 0.0;
    out[1] = r.value.origin.x; out[2] = r.value.origin.y; out[3] = r.value.origin.z;
    out[4] = r.value.direction.x; out[5] = r.value.direction.y; out[6] = r.value.direction.z;
}

// ---------------------------------------------------------------- host entry points
int ref_construct_triangle_bvh(const PtTriangle* triangles, int count, PtBvhNode* nodes) {
    const BVH bvh = construct_triangle_bvh(reinterpret_cast<const Triangle*>(triangles), count);
    std::memcpy(nodes, bvh.data(), bvh.size() * sizeof(Node));
    return static_cast<int>(bvh.size());
}

int ref_construct_sphere_bvh(const PtSphere* spheres, int count, PtBvhNode* nodes) {
    const BVH bvh = construct_sphere_bvh(reinterpret_cast<const Sphere*>(spheres), count);
    std::memcpy(nodes, bvh.data(), bvh.size() * sizeof(Node));
    return static_cast<int>(bvh.size());
}

void ref_rotation_matrix(double angle, double ax, double ay, double az, PtMat3* out) {
    const Mat3 m = rotation_matrix(angle, ax, ay, az);
    std::memcpy(out, &m, sizeof(m));
}

void ref_look_at_matrix(const PtVec3* position, const PtVec3* target, PtMat3* out) {
    const Mat3 m = look_at_matrix(Vec3{position->x, position->y, position->z}, Vec3{target->x, target->y, target->z});
    std::memcpy(out, &m, sizeof(m));
}

// Camera set-up as each built-in scene does it (e.g. src/main.cpp:833-840).
void ref_camera_look_at(const PtVec3* position, const PtVec3* target, double fov_y, double aperture, double focus_distance, PtCamera* out) {
    const Vec3 p{position->x, position->y, position->z};
    Camera camera = {};
    camera.target = Vec3{target->x, target->y, target->z};
    camera.orientation = look_at_matrix(p, camera.target);
    camera.distance = magnitude(p - camera.target);
    camera.fov_y = fov_y;
    camera.aperture = aperture;
    camera.focus_distance = (focus_distance > 0.0) ? focus_distance : camera.distance;
    std::memcpy(out, &camera, sizeof(camera));
}

// src/main.cpp:928-930, 987-994
void ref_camera_frame(const PtCamera* cam, double aspect_ratio, int width, int height, PtFrame* frame) {
    Camera camera;
    std::memcpy(&camera, cam, sizeof(camera));
    const Vec3 camera_x = get_column(camera.orientation, 0);
    const Vec3 camera_y = get_column(camera.orientation, 1);
    const Vec3 camera_z = get_column(camera.orientation, 2);

    const Vec3 camera_position = get_position(camera);

    const real viewport_height = 2.0f * std::tan(0.5f * camera.fov_y);
    const real viewport_width = aspect_ratio * viewport_height;

    const Vec3 step_x = camera.focus_distance * viewport_width * camera_x;
    const Vec3 step_y = camera.focus_distance * viewport_height * camera_y;
    const Vec3 bottom_left = camera_position - 0.5f * step_x - 0.5f * step_y - camera.focus_distance * camera_z;

    std::memset(frame, 0, sizeof(*frame));
    frame->width = width; frame->height = height;
    frame->row_begin = 0; frame->row_end = height;
    frame->sample_begin = 0; frame->sample_count = 1;
    frame->shard_index = 0; frame->shard_count = 1;
    frame->max_bounces = 50;
    frame->aperture = camera.aperture;
    std::memcpy(&frame->camera_position, &camera_position, 24);
    std::memcpy(&frame->camera_x, &camera_x, 24);
    std::memcpy(&frame->camera_y, &camera_y, 24);
    std::memcpy(&frame->bottom_left, &bottom_left, 24);
    std::memcpy(&frame->step_x, &step_x, 24);
    std::memcpy(&frame->step_y, &step_y, 24);
}

// ---------------------------------------------------------------- scenes
// Generic: caller-provided arrays; the BVHs are built by the reference's own builders.
RefScene* ref_scene_create(const PtSceneDesc* d) {
    RefScene* s = new RefScene();
    s->materials.resize(d->material_count);
    std::memcpy(s->materials.data(), d->materials, sizeof(Material) * d->material_count);
    if (d->sphere_count > 0) {
        s->spheres.resize(d->sphere_count);
        std::memcpy(s->spheres.data(), d->spheres, sizeof(Sphere) * d->sphere_count);
        s->sphere_material_indices.assign(d->sphere_material_indices, d->sphere_material_indices + d->sphere_count);
    }
    if (d->triangle_count > 0) {
        s->triangles.resize(d->triangle_count);
        std::memcpy(s->triangles.data(), d->triangles, sizeof(Triangle) * d->triangle_count);
        s->triangle_material_indices.assign(d->triangle_material_indices, d->triangle_material_indices + d->triangle_count);
    }
    Colour a, b;
    std::memcpy(&a, &d->background_gradient_start, 24);
    std::memcpy(&b, &d->background_gradient_end, 24);
    finish_scene(s, a, b);
    return s;
}

// The reference's active scene (src/main.cpp:800-840) with the model path as a parameter.
// load_triangles_file (src/model_loading.cpp:10-39) is restated with fopen/fread.
RefScene* ref_model_scene(const char* triangles_path, PtCamera* camera_out, double* aspect_out) {
    FILE* file = std::fopen(triangles_path, "rb");
    if (file == nullptr) return nullptr;
    std::fseek(file, 0, SEEK_END);
    const long file_size = std::ftell(file);
    std::fseek(file, 0, SEEK_SET);
    assert(file_size % sizeof(Triangle) == 0);                                   // model_loading.cpp:26
    std::vector<Triangle> model_triangles(file_size / sizeof(Triangle));         // :28
    const size_t got = std::fread(model_triangles.data(), 1, file_size, file);   // :31
    std::fclose(file);
    if (got != static_cast<size_t>(file_size)) return nullptr;

    RefScene* s = new RefScene();
    s->materials = {
        construct_lambertian_material(Colour{6.0f / 255.0f, 4.0f / 255.0f, 3.0f / 255.0f}),   // main.cpp:801
        construct_diffuse_light_material(Colour{1.0f, 1.0f, 1.0f}, 10.0f)                     // :802
    };
    s->triangle_material_indices.assign(model_triangles.size(), 0);                           // :806

    const Mat3 model_transform = rotation_matrix(-PI / 2.0f, 1.0f, 0.0f, 0.0f);               // :808
    for (Triangle& triangle : model_triangles) {                                              // :809-813
        triangle.a = model_transform * triangle.a;
        triangle.b = model_transform * triangle.b;
        triangle.c = model_transform * triangle.c;
    }
    s->triangles = std::move(model_triangles);

    s->spheres = {Sphere{Vec3{20.0f, 80.0f, 10.0f}, 20.0f}};                                  // :817
    s->sphere_material_indices = {1};                                                         // :818
    finish_scene(s, Colour{0.01f, 0.01f, 0.01f}, Colour{0.01f, 0.01f, 0.01f});                // :829-830

    if (camera_out != nullptr) {
        const Vec3 model_camera_start_position{0.0f, 150.0f, 150.0f};                         // :833
        Camera model_camera = {};
        model_camera.target = Vec3{0.0f, 0.0f, 0.0f};
        model_camera.orientation = look_at_matrix(model_camera_start_position, model_camera.target);
        model_camera.distance = magnitude(model_camera_start_position - model_camera.target);
        model_camera.fov_y = degrees_to_radians(40.0f);
        model_camera.aperture = 0.1f;
        model_camera.focus_distance = model_camera.distance;
        std::memcpy(camera_out, &model_camera, sizeof(model_camera));
    }
    if (aspect_out != nullptr) *aspect_out = 1.0f;                                            // :35-37
    return s;
}

// The reference's two other built-in scenes, restated from src/main.cpp (which cannot be compiled
// here) with the reference's own constructors, hashing and matrix code.  kind 0 = random spheres
// (src/main.cpp:516-625), kind 1 = Cornell box (src/main.cpp:627-798).
RefScene* ref_builtin_scene(int kind, PtCamera* camera_out, double* aspect_out) {
    RefScene* s = new RefScene();
    Camera camera = {};
    Colour bg_start{}, bg_end{};
    if (kind == 0) {
        static constexpr int SPHERE_COUNT = 22 * 22 + 4;                                            // :516
        s->materials.resize(SPHERE_COUNT);
        s->spheres.resize(SPHERE_COUNT);
        s->sphere_material_indices.resize(SPHERE_COUNT);
        Material* materials = s->materials.data();
        Sphere* spheres = s->spheres.data();
        int* sphere_material_indices = s->sphere_material_indices.data();

        int sphere_index = 0;
        materials[sphere_index] = construct_lambertian_material(Colour{0.5f, 0.5f, 0.5f});         // :522
        spheres[sphere_index] = Sphere{Vec3{0.0f, -1000.0f, 0.0f}, 1000.0f};
        sphere_material_indices[sphere_index] = sphere_index;
        ++sphere_index;

        u32 rng = 479001599;                                                                        // :527
        for (int a = -11; a < 11; ++a) {
            for (int b = -11; b < 11; ++b) {
                rng = random_number(rng);
                const real material_choice = real_from_rng(rng);
                rng = random_number(rng);
                const real x_offset = 0.9f * real_from_rng(rng);
                rng = random_number(rng);
                const real z_offset = 0.9f * real_from_rng(rng);
                const Vec3 sphere_centre{static_cast<real>(a) + x_offset, 0.2f, static_cast<real>(b) + z_offset};
                spheres[sphere_index] = Sphere{sphere_centre, 0.2f};
                if (material_choice < 0.8f) {                                                       // :542
                    rng = random_number(rng); const real colour_1_r = real_from_rng(rng);
                    rng = random_number(rng); const real colour_1_g = real_from_rng(rng);
                    rng = random_number(rng); const real colour_1_b = real_from_rng(rng);
                    const Colour colour_1{colour_1_r, colour_1_g, colour_1_b};
                    rng = random_number(rng); const real colour_2_r = real_from_rng(rng);
                    rng = random_number(rng); const real colour_2_g = real_from_rng(rng);
                    rng = random_number(rng); const real colour_2_b = real_from_rng(rng);
                    const Colour colour_2{colour_2_r, colour_2_g, colour_2_b};
                    const Colour albedo = colour_1 * colour_2;
                    materials[sphere_index] = construct_lambertian_material(albedo);
                } else if (material_choice < 0.95f) {                                               // :563
                    rng = random_number(rng); const real colour_r = 0.5f * real_from_rng(rng) + 0.5f;
                    rng = random_number(rng); const real colour_g = 0.5f * real_from_rng(rng) + 0.5f;
                    rng = random_number(rng); const real colour_b = 0.5f * real_from_rng(rng) + 0.5f;
                    const Colour albedo{colour_r, colour_g, colour_b};
                    rng = random_number(rng);
                    const real fuzziness = real_from_rng(rng);
                    materials[sphere_index] = construct_metal_material(albedo, fuzziness);
                } else {
                    materials[sphere_index] = construct_dielectric_material(1.5f);
                }
                sphere_material_indices[sphere_index] = sphere_index;
                ++sphere_index;
            }
        }
        assert(sphere_index == SPHERE_COUNT - 3);                                                   // :586
        materials[sphere_index] = construct_dielectric_material(1.5f);
        spheres[sphere_index] = Sphere{Vec3{0.0f, 1.0f, 0.0f}, 1.0f};
        sphere_material_indices[sphere_index] = sphere_index;
        ++sphere_index;
        materials[sphere_index] = construct_lambertian_material(Colour{0.4f, 0.2f, 0.1f});
        spheres[sphere_index] = Sphere{Vec3{-4.0f, 1.0f, 0.0f}, 1.0f};
        sphere_material_indices[sphere_index] = sphere_index;
        ++sphere_index;
        materials[sphere_index] = construct_metal_material(Colour{0.7f, 0.6f, 0.5f}, 0.0f);
        spheres[sphere_index] = Sphere{Vec3{4.0f, 1.0f, 0.0f}, 1.0f};
        sphere_material_indices[sphere_index] = sphere_index;
        ++sphere_index;
        assert(sphere_index == SPHERE_COUNT);                                                       // :603
        bg_start = Colour{1.0f, 1.0f, 1.0f};                                                        // :614-615
        bg_end = Colour{0.5f, 0.7f, 1.0f};

        const Vec3 sphere_camera_start_position{13.0f, 2.0f, 3.0f};                                 // :618-625
        camera.target = Vec3{0.0f, 0.0f, 0.0f};
        camera.orientation = look_at_matrix(sphere_camera_start_position, camera.target);
        camera.distance = magnitude(sphere_camera_start_position - camera.target);
        camera.fov_y = degrees_to_radians(20.0f);
        camera.aperture = 0.1f;
        camera.focus_distance = 10.0f;
        if (aspect_out != nullptr) *aspect_out = 3.0f / 2.0f;                                       // :33
    } else {
        s->materials = {                                                                            // :627-633
            construct_lambertian_material(Colour{0.65f, 0.05f, 0.05f}),
            construct_lambertian_material(Colour{0.73f, 0.73f, 0.73f}),
            construct_lambertian_material(Colour{0.12f, 0.45f, 0.15f}),
            construct_diffuse_light_material(Colour{1.0f, 1.0f, 1.0f}, 15.0f),
            construct_dielectric_material(1.5f)
        };
        s->spheres = {Sphere{Vec3{183.0f, 240.0f, 169.0f}, 75.0f}};                                 // :635-637
        s->sphere_material_indices = {4};                                                           // :639

        const Vec3 unit_box_vertices[] = {                                                          // :643-667
            Vec3{0.0f, 1.0f, 1.0f}, Vec3{0.0f, 0.0f, 1.0f}, Vec3{1.0f, 0.0f, 1.0f},
            Vec3{1.0f, 0.0f, 1.0f}, Vec3{1.0f, 1.0f, 1.0f}, Vec3{0.0f, 1.0f, 1.0f},
            Vec3{1.0f, 1.0f, 1.0f}, Vec3{1.0f, 0.0f, 1.0f}, Vec3{1.0f, 0.0f, 0.0f},
            Vec3{1.0f, 0.0f, 0.0f}, Vec3{1.0f, 1.0f, 0.0f}, Vec3{1.0f, 1.0f, 1.0f},
            Vec3{1.0f, 1.0f, 0.0f}, Vec3{1.0f, 0.0f, 0.0f}, Vec3{0.0f, 0.0f, 0.0f},
            Vec3{0.0f, 0.0f, 0.0f}, Vec3{0.0f, 1.0f, 0.0f}, Vec3{1.0f, 1.0f, 0.0f},
            Vec3{0.0f, 1.0f, 0.0f}, Vec3{0.0f, 0.0f, 0.0f}, Vec3{0.0f, 0.0f, 1.0f},
            Vec3{0.0f, 0.0f, 1.0f}, Vec3{0.0f, 1.0f, 1.0f}, Vec3{0.0f, 1.0f, 0.0f},
            Vec3{1.0f, 1.0f, 1.0f}, Vec3{1.0f, 1.0f, 0.0f}, Vec3{0.0f, 1.0f, 0.0f},
            Vec3{0.0f, 1.0f, 0.0f}, Vec3{0.0f, 1.0f, 1.0f}, Vec3{1.0f, 1.0f, 1.0f},
            Vec3{0.0f, 0.0f, 1.0f}, Vec3{0.0f, 0.0f, 0.0f}, Vec3{1.0f, 0.0f, 0.0f},
            Vec3{1.0f, 0.0f, 0.0f}, Vec3{1.0f, 0.0f, 1.0f}, Vec3{0.0f, 0.0f, 1.0f}
        };
        const Mat3 right_box_transform = scaling_matrix(165.0f, 165.0f, 165.0f) * rotation_matrix(-PI / 10.0f, 0.0f, 1.0f, 0.0f);   // :669
        Vec3 right_box_vertices[36] = {};
        for (int i = 0; i < 36; ++i) right_box_vertices[i] = right_box_transform * unit_box_vertices[i] + Vec3{130.0f, 0.0f, 65.0f};
        const Mat3 left_box_transform = scaling_matrix(165.0f, 330.0f, 165.0f) * rotation_matrix(PI / 12.0f, 0.0f, 1.0f, 0.0f);      // :679
        Vec3 left_box_vertices[36] = {};
        for (int i = 0; i < 36; ++i) left_box_vertices[i] = left_box_transform * unit_box_vertices[i] + Vec3{265.0f, 0.0f, 295.0f};

        s->triangles = {                                                                            // :687-749
            Triangle{Vec3{555.0f, 0.0f, 0.0f}, Vec3{555.0f, 0.0f, 555.0f}, Vec3{555.0f, 555.0f, 555.0f}},
            Triangle{Vec3{555.0f, 555.0f, 555.0f}, Vec3{555.0f, 555.0f, 0.0f}, Vec3{555.0f, 0.0f, 0.0f}},
            Triangle{Vec3{0.0f, 0.0f, 0.0f}, Vec3{0.0f, 555.0f, 0.0f}, Vec3{0.0f, 555.0f, 555.0f}},
            Triangle{Vec3{0.0f, 555.0f, 555.0f}, Vec3{0.0f, 0.0f, 555.0f}, Vec3{0.0f, 0.0f, 0.0f}},
            Triangle{Vec3{0.0f, 0.0f, 0.0f}, Vec3{0.0f, 0.0f, 555.0f}, Vec3{555.0f, 0.0f, 555.0f}},
            Triangle{Vec3{555.0f, 0.0f, 555.0f}, Vec3{555.0f, 0.0f, 0.0f}, Vec3{0.0f, 0.0f, 0.0f}},
            Triangle{Vec3{0.0f, 555.0f, 0.0f}, Vec3{555.0f, 555.0f, 0.0f}, Vec3{555.0f, 555.0f, 555.0f}},
            Triangle{Vec3{555.0f, 555.0f, 555.0f}, Vec3{0.0f, 555.0f, 555.0f}, Vec3{0.0f, 555.0f, 0.0f}},
            Triangle{Vec3{0.0f, 0.0f, 555.0f}, Vec3{0.0f, 555.0f, 555.0f}, Vec3{555.0f, 555.0f, 555.0f}},
            Triangle{Vec3{555.0f, 555.0f, 555.0f}, Vec3{555.0f, 0.0f, 555.0f}, Vec3{0.0f, 0.0f, 555.0f}},
            Triangle{Vec3{213.0f, 554.0f, 227.0f}, Vec3{343.0f, 554.0f, 227.0f}, Vec3{343.0f, 554.0f, 332.0f}},
            Triangle{Vec3{343.0f, 554.0f, 332.0f}, Vec3{213.0f, 554.0f, 332.0f}, Vec3{213.0f, 554.0f, 227.0f}}
        };
        for (int i = 0; i < 36; i += 3) s->triangles.push_back(Triangle{right_box_vertices[i], right_box_vertices[i + 1], right_box_vertices[i + 2]});
        for (int i = 0; i < 36; i += 3) s->triangles.push_back(Triangle{left_box_vertices[i], left_box_vertices[i + 1], left_box_vertices[i + 2]});
        s->triangle_material_indices = {2, 2, 0, 0, 1, 1, 1, 1, 1, 1, 3, 3,                         // :751-775
                                        1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                        1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
        bg_start = Colour{0.0f, 0.0f, 0.0f};                                                        // :787-788
        bg_end = Colour{0.0f, 0.0f, 0.0f};

        const Vec3 cornell_camera_start_position{278.0f, 278.0f, -800.0f};                          // :791-798
        camera.target = Vec3{278.0f, 278.0f, 0.0f};
        camera.orientation = look_at_matrix(cornell_camera_start_position, camera.target);
        camera.distance = magnitude(cornell_camera_start_position - camera.target);
        camera.fov_y = degrees_to_radians(40.0f);
        camera.aperture = 0.1f;
        camera.focus_distance = camera.distance;
        if (aspect_out != nullptr) *aspect_out = 1.0f;                                              // :34
    }
    finish_scene(s, bg_start, bg_end);
    if (camera_out != nullptr) std::memcpy(camera_out, &camera, sizeof(camera));
    return s;
}

void ref_scene_destroy(RefScene* s) { delete s; }

int ref_scene_counts(const RefScene* s, int* out /* materials, spheres, sphere nodes, triangles, triangle nodes */) {
    out[0] = static_cast<int>(s->materials.size());
    out[1] = static_cast<int>(s->spheres.size());
    out[2] = static_cast<int>(s->sphere_bvh.size());
    out[3] = static_cast<int>(s->triangles.size());
    out[4] = static_cast<int>(s->triangle_bvh.size());
    return 0;
}

// Copies the scene arrays out (any pointer may be null).
int ref_scene_export(const RefScene* s, PtMaterial* materials, PtSphere* spheres, int* sphere_mat, PtBvhNode* sphere_nodes,
                     PtTriangle* triangles, int* triangle_mat, PtBvhNode* triangle_nodes, PtColour* background /* [2] */) {
    if (materials) std::memcpy(materials, s->materials.data(), s->materials.size() * sizeof(Material));
    if (spheres) std::memcpy(spheres, s->spheres.data(), s->spheres.size() * sizeof(Sphere));
    if (sphere_mat) std::memcpy(sphere_mat, s->sphere_material_indices.data(), s->sphere_material_indices.size() * sizeof(int));
    if (sphere_nodes) std::memcpy(sphere_nodes, s->sphere_bvh.data(), s->sphere_bvh.size() * sizeof(Node));
    if (triangles) std::memcpy(triangles, s->triangles.data(), s->triangles.size() * sizeof(Triangle));
    if (triangle_mat) std::memcpy(triangle_mat, s->triangle_material_indices.data(), s->triangle_material_indices.size() * sizeof(int));
    if (triangle_nodes) std::memcpy(triangle_nodes, s->triangle_bvh.data(), s->triangle_bvh.size() * sizeof(Node));
    if (background) {
        std::memcpy(&background[0], &s->scene.background_gradient_start, 24);
        std::memcpy(&background[1], &s->scene.background_gradient_end, 24);
    }
    return 0;
}

// ---------------------------------------------------------------- the hot path
// Rows are claimed from a shared counter by `threads` workers: the portable twin of the
// reference's row work queue (src/main.cpp:41-72, 125-177, 1001-1022).  max_bounces must be
// 50: MAX_BOUNCE_COUNT is a compile-time constant of the reference (src/path_tracing.cpp:172).
int ref_render(const RefScene* s, const PtFrame* f, double* accum, int threads) {
    if (f->max_bounces != 50) return 1;
    std::atomic<int> next_row{f->row_begin};
    auto worker = [&]() {
        for (;;) {
            const int row = next_row.fetch_add(1);
            if (row >= f->row_end) break;
            if (f->shard_count > 1 && ((row - f->row_begin) / PT_ROW_BLOCK) % f->shard_count != f->shard_index) continue;
            render_row(*s, *f, row, accum);
        }
    };
    if (threads <= 1) {
        worker();
    } else {
        std::vector<std::thread> pool;
        for (int i = 0; i < threads; ++i) pool.emplace_back(worker);
        for (std::thread& t : pool) t.join();
    }
    return 0;
}

// Colour of one sample per pixel (r, g, b), no accumulation: for locating a divergence.
int ref_sample_colours(const RefScene* s, const PtFrame* f, int sample, double* rgb) {
    if (f->max_bounces != 50) return 1;
    for (int row = f->row_begin; row < f->row_end; ++row) {
        for (int column = 0; column < f->width; ++column) {
            const Ray ray = primary_ray(*f, row, column, sample);
            const Colour c = intersect(ray, s->scene);
            double* out = rgb + 3ll * (static_cast<long long>(row) * f->width + column);
            out[0] = c.r; out[1] = c.g; out[2] = c.b;
        }
    }
    return 0;
}

int ref_primary_rays(const PtFrame* f, int sample, double* rays /* 6 per pixel */) {
    for (int row = f->row_begin; row < f->row_end; ++row) {
        for (int column = 0; column < f->width; ++column) {
            const Ray ray = primary_ray(*f, row, column, sample);
            std::memcpy(rays + 6ll * (static_cast<long long>(row) * f->width + column), &ray, sizeof(ray));
        }
    }
    return 0;
}

// First-bounce indices of both traversals (src/path_tracing.cpp:177-178).
int ref_primary_hits(const RefScene* s, const PtFrame* f, int* triangle_index, int* sphere_index) {
    for (int row = f->row_begin; row < f->row_end; ++row) {
        for (int column = 0; column < f->width; ++column) {
            const Ray ray = primary_ray(*f, row, column, f->sample_begin);
            const ClosestShapeIntersection cs = (s->scene.sphere_bvh != nullptr) ? intersect(ray, *s->scene.sphere_bvh, 0, s->scene.spheres) : MISS;
            const ClosestShapeIntersection ct = (s->scene.triangle_bvh != nullptr) ? intersect(ray, *s->scene.triangle_bvh, 0, s->scene.triangles) : MISS;
            const long long i = static_cast<long long>(row) * f->width + column;
            if (triangle_index) triangle_index[i] = ct.index;
            if (sphere_index) sphere_index[i] = cs.index;
        }
    }
    return 0;
}

// Closest hit of arbitrary rays against the triangle BVH: out index and distance.
int ref_trace_triangles(const RefScene* s, const double* rays, long long n, int* index, double* distance) {
    for (long long i = 0; i < n; ++i) {
        const ClosestShapeIntersection c = (s->scene.triangle_bvh != nullptr) ? intersect(ray_from(rays + 6 * i), *s->scene.triangle_bvh, 0, s->scene.triangles) : MISS;
        index[i] = c.index; distance[i] = c.distance;
    }
    return 0;
}

// src/main.cpp:1024-1034 with sample_count standing for (sample + 1)
int ref_tonemap_bgra8(const double* accum, long long pixel_count, int sample_count, unsigned char* out) {
    for (long long pixel_index = 0; pixel_index < pixel_count; ++pixel_index) {
        const long long index = 4 * pixel_index;
        const real b = std::min<real>(std::sqrt(accum[index + 0] / static_cast<real>(sample_count)), 1.0f);
        const real g = std::min<real>(std::sqrt(accum[index + 1] / static_cast<real>(sample_count)), 1.0f);
        const real r = std::min<real>(std::sqrt(accum[index + 2] / static_cast<real>(sample_count)), 1.0f);
        out[index + 0] = static_cast<unsigned char>(255.0f * b);
        out[index + 1] = static_cast<unsigned char>(255.0f * g);
        out[index + 2] = static_cast<unsigned char>(255.0f * r);
        out[index + 3] = 255;
    }
    return 0;
}

double ref_degrees_to_radians(double degrees) { return degrees_to_radians(degrees); }

// The camera controls of the frame loop, src/main.cpp:890-985, with the reference's own matrix and vector code; the
// Win32 input structs (MouseInput / KeyboardInput, src/main.cpp:214-260) are replaced by PtViewerInput.  Returns camera_modified.
int ref_camera_apply_input(PtCamera* cam, const PtViewerInput* in) {
    Camera camera;
    std::memcpy(&camera, cam, sizeof(camera));
    const real frame_duration = in->frame_duration;
    bool camera_modified = false;
    const bool left_mouse_button_held = in->left_button_held != 0;
    const int mouse_delta_x = in->mouse_delta_x, mouse_delta_y = in->mouse_delta_y;
    if (left_mouse_button_held && (mouse_delta_x != 0 || mouse_delta_y != 0)) {                     // :898-908
        static constexpr real MOUSE_SENSITIVITY = 4.0f;
        const real camera_yaw_delta = static_cast<real>(mouse_delta_x) * MOUSE_SENSITIVITY * frame_duration;
        const real camera_pitch_delta = -static_cast<real>(mouse_delta_y) * MOUSE_SENSITIVITY * frame_duration;
        const Mat3 camera_yaw_delta_rotation = rotation_matrix(degrees_to_radians(camera_yaw_delta), 0.0f, 0.0f, 1.0f);
        const Mat3 camera_pitch_delta_rotation = rotation_matrix(degrees_to_radians(camera_pitch_delta), 1.0f, 0.0f, 0.0f);
        camera.orientation = camera_yaw_delta_rotation * camera.orientation * camera_pitch_delta_rotation;
        camera_modified = true;
    }
    static constexpr float SCROLL_SENSITIVITY = 0.95f;                                              // :910-924
    if (in->scroll > 0) {
        for (int i = 0; i < in->scroll; ++i) camera.distance *= SCROLL_SENSITIVITY;
        camera_modified = true;
    } else if (in->scroll < 0) {
        for (int i = in->scroll; i < 0; ++i) camera.distance /= SCROLL_SENSITIVITY;
        camera_modified = true;
    }
    const Vec3 camera_x = get_column(camera.orientation, 0);                                        // :928-929
    const Vec3 camera_y = get_column(camera.orientation, 1);
    static constexpr float CAMERA_SPEED = 0.6f;                                                     // :932-952
    if (in->key_a) { camera.target = camera.target - camera.distance * CAMERA_SPEED * frame_duration * camera_x; camera_modified = true; }
    if (in->key_d) { camera.target = camera.target + camera.distance * CAMERA_SPEED * frame_duration * camera_x; camera_modified = true; }
    if (in->key_w) { camera.target = camera.target + camera.distance * CAMERA_SPEED * frame_duration * camera_y; camera_modified = true; }
    if (in->key_s) { camera.target = camera.target - camera.distance * CAMERA_SPEED * frame_duration * camera_y; camera_modified = true; }
    static constexpr real FOV_SENSITIVITY = 0.1f;                                                   // :954-963
    if (in->key_q) { camera.fov_y -= FOV_SENSITIVITY * frame_duration; camera_modified = true; }
    if (in->key_e) { camera.fov_y += FOV_SENSITIVITY * frame_duration; camera_modified = true; }
    static constexpr real FOCUS_SENSITIVITY = 1.0f;                                                 // :965-974
    if (in->key_up) { camera.focus_distance += FOCUS_SENSITIVITY * frame_duration * camera.distance; camera_modified = true; }
    if (in->key_down) { camera.focus_distance -= FOCUS_SENSITIVITY * frame_duration * camera.distance; camera_modified = true; }
    static constexpr real APERTURE_SENSITIVITY = 0.1f;                                              // :976-985
    if (in->key_left) { camera.aperture -= APERTURE_SENSITIVITY * frame_duration; camera_modified = true; }
    if (in->key_right) { camera.aperture += APERTURE_SENSITIVITY * frame_duration; camera_modified = true; }
    std::memcpy(cam, &camera, sizeof(camera));
    return camera_modified ? 1 : 0;
}

}  // extern "C"
