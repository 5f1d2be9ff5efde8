/* A C (not C++) caller of include/pathtracer_amd.h: what a maintainer of the reference renderer adds to
 * src/main.cpp to hand its row work queue (src/main.cpp:1001-1022) to the GPU.  The scene is the reference's
 * active one, set up from its own constants (src/main.cpp:800-840) into arrays this program owns, in the
 * reference's record layouts; the library gets pointers to them, exactly as it would get the reference's.
 *
 *   render_rook <rook.triangles> <width> <height> <samples> <out.f64>
 *
 * writes the accumulation buffer (4 * width * height doubles, BGRA + count) to <out.f64>. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pathtracer_amd.h"

static const float PI = 3.14159265358979f;                                /* src/main.cpp:27 */

static void die(const char* what) {
    fprintf(stderr, "render_rook: %s: %s\n", what, pt_last_error());
    exit(1);
}

static PtVec3 transform(const PtMat3* m, PtVec3 v) {                      /* Mat3 * Vec3, src/linear_algebra.cpp:42-48 */
    PtVec3 r;
    r.x = m->rows[0][0] * v.x + m->rows[0][1] * v.y + m->rows[0][2] * v.z;
    r.y = m->rows[1][0] * v.x + m->rows[1][1] * v.y + m->rows[1][2] * v.z;
    r.z = m->rows[2][0] * v.x + m->rows[2][1] * v.y + m->rows[2][2] * v.z;
    return r;
}

int main(int argc, char** argv) {
    if (argc != 6) { fprintf(stderr, "usage: render_rook <rook.triangles> <width> <height> <samples> <out.f64>\n"); return 2; }
    const int width = atoi(argv[2]), height = atoi(argv[3]), samples = atoi(argv[4]);

    /* src/main.cpp:800-819: materials, model (rotated from z-up to y-up), its BVH, the light and its BVH */
    PtMaterial materials[2];
    memset(materials, 0, sizeof materials);
    materials[0].type = PT_LAMBERTIAN;
    materials[0].v[0] = 6.0f / 255.0f; materials[0].v[1] = 4.0f / 255.0f; materials[0].v[2] = 3.0f / 255.0f;
    materials[1].type = PT_DIFFUSE_LIGHT;
    materials[1].v[0] = 1.0f; materials[1].v[1] = 1.0f; materials[1].v[2] = 1.0f; materials[1].v[3] = 10.0f;

    PtTriangle* triangles = NULL;
    int64_t triangle_count = 0;
    if (pt_load_triangles_file(argv[1], &triangles, &triangle_count) != PT_OK) die("pt_load_triangles_file");
    PtMat3 model_transform;
    if (pt_rotation_matrix(-PI / 2.0f, 1.0f, 0.0f, 0.0f, &model_transform) != PT_OK) die("pt_rotation_matrix");
    for (int64_t i = 0; i < triangle_count; ++i) {
        triangles[i].a = transform(&model_transform, triangles[i].a);
        triangles[i].b = transform(&model_transform, triangles[i].b);
        triangles[i].c = transform(&model_transform, triangles[i].c);
    }
    int32_t* triangle_materials = calloc((size_t)triangle_count, sizeof(int32_t));
    PtBvhNode* triangle_bvh = malloc((size_t)(2 * triangle_count - 1) * sizeof(PtBvhNode));
    if (pt_construct_triangle_bvh(triangles, (int32_t)triangle_count, triangle_bvh) != PT_OK) die("pt_construct_triangle_bvh");
    const PtSphere light = {{20.0f, 80.0f, 10.0f}, 20.0f};
    const int32_t light_material = 1;
    PtBvhNode light_bvh[1];
    if (pt_construct_sphere_bvh(&light, 1, light_bvh) != PT_OK) die("pt_construct_sphere_bvh");

    /* src/main.cpp:821-831: the Scene, plus the counts it never stores */
    PtSceneDesc desc;
    memset(&desc, 0, sizeof desc);
    desc.materials = materials;               desc.material_count = 2;
    desc.spheres = &light;                    desc.sphere_count = 1;
    desc.sphere_bvh = light_bvh;              desc.sphere_bvh_node_count = 1;
    desc.sphere_material_indices = &light_material;
    desc.triangles = triangles;               desc.triangle_count = (int32_t)triangle_count;
    desc.triangle_bvh = triangle_bvh;         desc.triangle_bvh_node_count = (int32_t)(2 * triangle_count - 1);
    desc.triangle_material_indices = triangle_materials;
    desc.background_gradient_start.r = desc.background_gradient_start.g = desc.background_gradient_start.b = 0.01f;
    desc.background_gradient_end = desc.background_gradient_start;
    PtScene* scene = NULL;
    if (pt_scene_create(&desc, 0, &scene) != PT_OK) die("pt_scene_create");

    /* src/main.cpp:833-840 and :987-994: the camera and this frame's ray basis */
    const PtVec3 start = {0.0f, 150.0f, 150.0f}, target = {0.0f, 0.0f, 0.0f};
    const double fov_y = 40.0 / 180.0f * PI;                               /* degrees_to_radians, src/main.cpp:28-30 */
    PtCamera camera;
    if (pt_camera_look_at(&start, &target, fov_y, 0.1f, 0.0, &camera) != PT_OK) die("pt_camera_look_at");
    PtFrame frame;
    if (pt_camera_frame(&camera, 1.0, width, height, &frame) != PT_OK) die("pt_camera_frame");   /* ASPECT_RATIO 1, src/main.cpp:33 */
    frame.sample_begin = 0;
    frame.sample_count = samples;              /* `samples` frames of the reference's loop in one call */

    /* src/main.cpp:1001-1022: every row of the frame; the buffer is the reference's `pixels` (src/main.cpp:865) */
    double* pixels = calloc((size_t)4 * width * height, sizeof(double));
    if (pt_render(scene, &frame, pixels) != PT_OK) die("pt_render");

    FILE* out = fopen(argv[5], "wb");
    if (out == NULL || fwrite(pixels, sizeof(double), (size_t)4 * width * height, out) != (size_t)4 * width * height) { perror(argv[5]); return 1; }
    fclose(out);
    pt_scene_destroy(scene);
    pt_free(triangles);
    free(triangle_materials); free(triangle_bvh); free(pixels);
    return 0;
}
