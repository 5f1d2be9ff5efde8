/*
 * pathtracer_amd — C ABI of the MI355X (gfx950) bounce-loop path.
 *
 * Every entry point is `extern "C"`, takes plain pointers and sizes and returns an
 * int status (PT_OK == 0).  The reference renderer (LewisBray/PathTracer) has no
 * plugin / FFI interface: all its functions are `static` in one translation unit
 * (src/main.cpp:11-18).  The seam this library replaces is the row work-queue
 * payload `RenderWorkQueue::Entry` (src/main.cpp:42-54) that is handed to
 * `render_scanline` (src/main.cpp:74-123), plus the host entry points that feed it.
 * Each declaration below cites the reference code it stands in for.
 *
 * All floating point is IEEE binary64 ("real" is double, src/types.h:9); all
 * records are binary compatible with the reference's structs so that a reference
 * build can pass its own arrays straight through.
 */
#ifndef PATHTRACER_AMD_H
#define PATHTRACER_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes (the reference only asserts; src/model_loading.cpp:21-36) ---- */
enum {
    PT_OK = 0,
    PT_ERR_INVALID_ARGUMENT = 1,
    PT_ERR_IO = 2,
    PT_ERR_DEVICE = 3,
    PT_ERR_OUT_OF_MEMORY = 4,
    PT_ERR_UNSUPPORTED = 5
};

/* ---- plain records, layout-identical to the reference's ---- */
typedef struct PtVec3 { double x, y, z; } PtVec3;                       /* src/linear_algebra.h:6-10  (24 B) */
typedef struct PtMat3 { double rows[3][3]; } PtMat3;                    /* src/linear_algebra.h:22-24 (72 B) */
typedef struct PtColour { double r, g, b; } PtColour;                   /* src/colour.h:6-10          (24 B) */
typedef struct PtRay { PtVec3 origin, direction; } PtRay;               /* src/geometry.h:9-12        (48 B) */
typedef struct PtAabb { PtVec3 min, max; } PtAabb;                      /* src/geometry.h:14-17       (48 B) */
typedef struct PtSphere { PtVec3 centre; double radius; } PtSphere;     /* src/geometry.h:28-31       (32 B) */
typedef struct PtTriangle { PtVec3 a, b, c; } PtTriangle;               /* src/geometry.h:42-46       (72 B) */

/* src/bvh.h:8-13.  A leaf has left == 0 && right == 0 and names its primitive in
 * `index`; the root is node 0; a tree over N primitives has 2N-1 nodes in pre-order. (64 B) */
typedef struct PtBvhNode {
    PtAabb aabb;
    int32_t index, left, right;
    int32_t pad_;
} PtBvhNode;

/* src/material.h:25-41: a 32-byte union followed by the type tag (40 B).
 *   LAMBERTIAN    v[0..2] = albedo
 *   METAL         v[0..2] = albedo, v[3] = fuzziness
 *   DIELECTRIC    v[0]    = refraction index
 *   DIFFUSE_LIGHT v[0..2] = emission colour, v[3] = emission power */
enum { PT_LAMBERTIAN = 0, PT_METAL = 1, PT_DIELECTRIC = 2, PT_DIFFUSE_LIGHT = 3 };
typedef struct PtMaterial {
    double v[4];
    int32_t type;
    int32_t pad_;
} PtMaterial;

/* src/path_tracing.h:27-35 (128 B) */
typedef struct PtCamera {
    PtMat3 orientation;
    PtVec3 target;
    double distance;
    double fov_y;
    double aperture;
    double focus_distance;
} PtCamera;

/* src/path_tracing.h:9-22 `Scene`, plus the element counts the reference never
 * stores.  All pointers are host arrays owned by the caller; a null BVH means
 * "no primitives of that kind" exactly as in src/path_tracing.cpp:177-178. */
typedef struct PtSceneDesc {
    const PtMaterial* materials;
    int32_t material_count;

    const PtSphere* spheres;
    int32_t sphere_count;
    const PtBvhNode* sphere_bvh;
    int32_t sphere_bvh_node_count;
    const int32_t* sphere_material_indices;

    const PtTriangle* triangles;
    int32_t triangle_count;
    const PtBvhNode* triangle_bvh;
    int32_t triangle_bvh_node_count;
    const int32_t* triangle_material_indices;

    PtColour background_gradient_start;
    PtColour background_gradient_end;
} PtSceneDesc;

/* `RenderWorkQueue::Entry` (src/main.cpp:42-54) generalised from one row and one
 * sample to a row range and a sample range, with the compile-time constants
 * CLIENT_WIDTH / CLIENT_HEIGHT (src/main.cpp:38-39) and MAX_BOUNCE_COUNT
 * (src/path_tracing.cpp:172) made run-time fields.  Row 0 is the bottom row. */
typedef struct PtFrame {
    int32_t width, height;
    int32_t row_begin, row_end;        /* rows [row_begin, row_end) are rendered      */
    int32_t sample_begin, sample_count;/* samples [sample_begin, sample_begin+count)  */
    int32_t max_bounces;               /* reference value: 50                          */
    int32_t reserved_;
    double aperture;
    PtVec3 camera_position, camera_x, camera_y;
    PtVec3 bottom_left, step_x, step_y;
    /* Sharding of the row range over several callers (one per GPU), the way the reference's
     * work queue deals rows to threads: rows are taken in blocks of PT_ROW_BLOCK starting at
     * row_begin, and block b is rendered only if b % shard_count == shard_index.
     * shard_count <= 1 renders every row. */
    int32_t shard_index, shard_count;
} PtFrame;

enum { PT_ROW_BLOCK = 8 };

typedef struct PtScene PtScene;          /* device-resident scene (opaque)  */
typedef struct PtHostScene PtHostScene;  /* host-side scene arrays (opaque) */

/* Text of the last error raised on the calling thread ("" if none). */
const char* pt_last_error(void);

/* ---------------- host entry points kept from the reference ---------------- */

/* load_triangles_file (src/model_loading.cpp:10-39): whole file = raw 72-byte
 * records.  *triangles is malloc'ed; release with pt_free. */
int pt_load_triangles_file(const char* filename, PtTriangle** triangles, int64_t* count);
/* save_triangles_file (src/model_loading.cpp:41-61). */
int pt_save_triangles_file(const char* filename, const PtTriangle* triangles, int64_t count);
/* parse_stl_file (src/model_loading.cpp:63-196): ASCII STL -> triangles (facet normals are
 * read and dropped, vertices converted with strtod).  The reference's asset converter; it
 * asserts on malformed input, this returns PT_ERR_IO.  *triangles is malloc'ed (null if the
 * file holds no facet); release with pt_free. */
int pt_parse_stl_file(const char* filename, PtTriangle** triangles, int64_t* count);
void pt_free(void* p);

/* construct_triangle_bvh / construct_sphere_bvh (src/bvh.cpp:81-100 / :60-79).
 * `nodes` must hold 2*count-1 records; the output is identical, node for node,
 * to the reference's std::vector<Node> when both are built with the same standard library:
 * the reference sorts with std::sort, which is unstable, so which of several primitives with
 * equal box minima goes left at a split is the library's choice (libstdc++ here). */
int pt_construct_triangle_bvh(const PtTriangle* triangles, int32_t count, PtBvhNode* nodes);
int pt_construct_sphere_bvh(const PtSphere* spheres, int32_t count, PtBvhNode* nodes);

/* rotation_matrix / look_at_matrix (src/linear_algebra.cpp:81-101 / :103-122). */
int pt_rotation_matrix(double angle, double axis_x, double axis_y, double axis_z, PtMat3* out);
int pt_look_at_matrix(const PtVec3* position, const PtVec3* target, PtMat3* out);
/* Camera set-up as the reference does it for each built-in scene (e.g.
 * src/main.cpp:833-840): orientation = look_at, distance = |position - target|;
 * focus_distance <= 0 selects "focus at the target" (focus = distance). */
int pt_camera_look_at(const PtVec3* position, const PtVec3* target, double fov_y,
                      double aperture, double focus_distance, PtCamera* out);
/* Per-frame camera basis (src/main.cpp:928-930, 987-994 and get_position,
 * src/path_tracing.cpp:229-231).  Fills every camera field of *frame plus
 * width/height; row range = whole image, sample range = {0,1}, max_bounces = 50. */
int pt_camera_frame(const PtCamera* camera, double aspect_ratio, int32_t width, int32_t height,
                    PtFrame* frame);

/* One frame's worth of the reference's mouse and keyboard state (MouseInput / KeyboardInput as the frame loop
 * reads them, src/main.cpp:892-985) and the seconds since the previous frame (:887). */
typedef struct PtViewerInput {
    int32_t left_button_held;            /* left button down in this frame and in the previous one (:895) */
    int32_t mouse_delta_x, mouse_delta_y;/* pointer movement since the previous frame (:896-897)          */
    int32_t scroll;                      /* wheel notches since the previous frame, either sign (:911-924) */
    int32_t key_a, key_d, key_w, key_s;  /* move the target left/right/up/down (:934-952)                  */
    int32_t key_q, key_e;                /* narrow / widen the field of view (:955-963)                    */
    int32_t key_up, key_down;            /* focus distance (:966-974)                                      */
    int32_t key_left, key_right;         /* aperture (:977-985)                                            */
    double frame_duration;
} PtViewerInput;
/* The camera controls of the reference's frame loop (src/main.cpp:890-985): orbit by mouse drag (yaw about z,
 * pitch about x, 4 degrees per pixel and second), zoom by wheel (distance * or / 0.95f per notch), target moved
 * along the camera's x / y axes, field of view, focus distance and aperture by keys.  *camera_modified is set to
 * 1 if anything changed (the reference then restarts its accumulation, :996-999), else 0. */
int pt_camera_apply_input(PtCamera* camera, const PtViewerInput* input, int32_t* camera_modified);

/* Built-in scenes.  PT_SCENE_MODEL is the reference's active scene
 * (src/main.cpp:800-840) with the model file chosen by the caller; the chess
 * scenes are this library's own definitions over the reference's six model
 * files (the reference defines no multi-piece scene). */
enum {
    PT_SCENE_MODEL = 0,        /* one model file, reference materials/light/camera            */
    PT_SCENE_CHESS6_ROW = 1,   /* pawn..king in a row, 83,504 triangles                       */
    PT_SCENE_CHESS32 = 2,      /* 32-piece starting position + board, 422,598 triangles       */
    PT_SCENE_CHESS32_MIXED = 3,/* same geometry, all four material types, sky-like background */
    PT_SCENE_CHESS_GRID = 4,   /* chess32 replicated on a grid_n x grid_n lattice of boards    */
    PT_SCENE_RANDOM_SPHERES = 5,/* src/main.cpp:516-625                                       */
    PT_SCENE_CORNELL_BOX = 6   /* src/main.cpp:627-798                                        */
};
/* `path` is a .triangles file for PT_SCENE_MODEL, a directory holding
 * pawn/rook/knight/bishop/queen/king.triangles for the chess scenes, ignored
 * otherwise.  `param` is grid_n for PT_SCENE_CHESS_GRID, ignored otherwise. */
int pt_host_scene_build(int32_t kind, const char* path, int32_t param, PtHostScene** out);
int pt_host_scene_desc(const PtHostScene* scene, PtSceneDesc* desc);
int pt_host_scene_camera(const PtHostScene* scene, PtCamera* camera, double* aspect_ratio);
void pt_host_scene_destroy(PtHostScene* scene);

/* Tone-map (src/main.cpp:1024-1034): out[4i+{0,1,2}] = (u8)(255 * min(sqrt(accum/sample_count), 1)),
 * out[4i+3] = 255; BGRA order like the accumulation buffer. */
int pt_tonemap_bgra8(const double* accum, int64_t pixel_count, int32_t sample_count, uint8_t* out);
/* write_pixel_data_to_file (src/main.cpp:179-212): raw RGBA8 (B and R swapped on
 * the way out), rows bottom-to-top, no header.  `bgra` is left unchanged. */
int pt_write_pixel_data_file(const char* filename, const uint8_t* bgra, int64_t pixel_count);
/* Convenience, not in the reference (its only image output is the headerless dump above):
 * the same BGRA8 image as a binary PPM (P6), top row first, so that a viewer can open it. */
int pt_write_ppm_file(const char* filename, const uint8_t* bgra, int32_t width, int32_t height);

/* ---------------- the GPU path ---------------- */

/* Copies the scene (src/path_tracing.h:9-22) into the HBM of `device` and lays
 * the BVH out for traversal.  Validates every index the reference would follow
 * blindly.  The caller keeps ownership of the arrays in *desc.
 * The trees in *desc are the reference's to the letter: results are those of the reference's
 * traversal of exactly these nodes.  A tree built by pt_construct_*_bvh over proper primitives has
 * boxes that nest (every inner box holds its children's) and is searched through a faster
 * hierarchy of the library's own, which reaches the same leaves.  A tree whose boxes do not nest
 * (a hollow sphere's inverted box, src/geometry.cpp:51-57, or a caller's own inner boxes) is
 * walked as it is, without skipping anything for being far: same bytes, lower speed. */
int pt_scene_create(const PtSceneDesc* desc, int32_t device, PtScene** out);
void pt_scene_destroy(PtScene* scene);

/* render_scanline (src/main.cpp:74-123) over frame->rows x frame->samples:
 * for every pixel, in ascending sample order,
 *   accum[4*(row*width+col) + {0,1,2,3}] += {b, g, r, 1.0}.
 * `accum` is a caller-owned host buffer of 4*width*height doubles. */
int pt_render(PtScene* scene, const PtFrame* frame, double* accum);
/* Same, with `accum_device` in the HBM of the scene's device and the work
 * enqueued on `hip_stream` (a hipStream_t; null = the default stream).  Returns
 * after the launch; the caller synchronises the stream.  A scene handle owns one work counter and
 * one colour workspace, so its launches run one after another: those on one stream in stream order, and a
 * launch on another stream (the null stream included: pt_render and pt_primary_hits use it) is put behind
 * the end of the handle's previous launch (hipStreamWaitEvent).  Frames meant to overlap take one handle each.
 * The workspace (per-path colour records: 32 bytes x pixels of the call's rows x samples, in batches when that
 * is more than the limit) is allocated on demand and kept by the handle until pt_scene_destroy or a lower
 * pt_scene_set_workspace_limit. */
int pt_render_device(PtScene* scene, const PtFrame* frame, double* accum_device, void* hip_stream);

/* Cap, in bytes, of the device memory the handle keeps for its per-path colour records (default 24 GiB; the
 * least a launch takes is one group of 16 samples of its rows, 0.5 KiB per pixel).  A smaller cap means more
 * sample batches per launch, never other bytes.  What the handle holds beyond a lowered cap is freed at once
 * (after its launches have finished).  Not in the reference (which keeps no per-sample state). */
int pt_scene_set_workspace_limit(PtScene* scene, uint64_t bytes);

/* One iteration of the reference's frame loop without the window (src/main.cpp:890-1036): the controls in
 * `input` (may be null: nothing pressed) are applied to *camera; if that changed it, `accum_device` (4*width*height
 * doubles in the HBM of the scene's device) is zeroed and *sample set to 0 (:996-999); every pixel's samples
 * *sample .. *sample + samples_per_frame - 1 are added (the reference adds one per frame, :1001-1022; more per call
 * give the bytes of as many frames with an unmoved camera); *sample is advanced (:1036).  pt_tonemap_device with
 * sample_count = *sample then gives the frame the reference would show (:1024-1034).  Work is enqueued on
 * `hip_stream`; the call returns after the launch. */
int pt_viewer_frame(PtScene* scene, PtCamera* camera, const PtViewerInput* input, double aspect_ratio, int32_t width, int32_t height,
                    int32_t samples_per_frame, int32_t* sample, double* accum_device, void* hip_stream);

/* pt_render_device for callers that cannot allocate device memory (a stream under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments, same
 * bytes in `accum_device`; the per-path colour records and the work counter live in a workspace that is
 * part of the library's code object (PT_NOALLOC_MAX_TILES * 32 KiB + 1 KiB of HBM on every GPU the library
 * is loaded on, held until the process ends), so the call makes no hipMalloc / hipFree.  Limit: the rows the
 * call renders (its shard of [row_begin, row_end)) may hold at most PT_NOALLOC_MAX_TILES tiles of 8x8 pixels
 * (524,288 pixels, e.g. 1920 x 272 rows); beyond it PT_ERR_UNSUPPORTED is returned and nothing is rendered
 * - split the frame into row ranges.  Any sample count is taken (16 samples per pixel pass through the
 * workspace at a time).  All calls on one GPU share that GPU's workspace, whatever scene handle they go
 * through: each call is put behind the end of the one before it on that GPU (an event of the library's own),
 * so calls from different handles or streams run one after another.  On a stream that is being captured the
 * call enqueues its kernels and memsets only (no event is recorded, queried or waited for, and pt_last_render_ms
 * has nothing to report): the graph orders what it holds, and replays of graphs that use the workspace must
 * not overlap each other or other calls on that GPU. */
enum { PT_NOALLOC_MAX_TILES = 8192 };
int pt_render_device_noalloc(PtScene* scene, const PtFrame* frame, double* accum_device, void* hip_stream);

/* The tone-map of src/main.cpp:1024-1034 on the GPU: `accum_device` (4*pixel_count
 * doubles) and `bgra8_device` (4*pixel_count bytes) are in the HBM of `device`, the
 * kernel is enqueued on `hip_stream`.  Same bytes as pt_tonemap_bgra8.  Lets a
 * multi-GPU job gather 4 bytes per pixel instead of 32. */
int pt_tonemap_device(int32_t device, const double* accum_device, int64_t pixel_count, int32_t sample_count,
                      uint8_t* bgra8_device, void* hip_stream);

/* First-bounce `ClosestShapeIntersection.index` of both traversals
 * (src/path_tracing.cpp:177-178) for sample frame->sample_begin, -1 = miss.
 * Host buffers of width*height int32 each; rows outside the frame's row range
 * are left untouched.  Either pointer may be null. */
int pt_primary_hits(PtScene* scene, const PtFrame* frame, int32_t* triangle_index, int32_t* sphere_index);

/* Kernel time in milliseconds of the most recent pt_render / pt_render_device on
 * this scene (pt_viewer_frame and pt_primary_hits launch too and count), measured with HIP events
 * on the launch stream (blocks until the launch has finished).  Not in the reference: measurement only. */
int pt_last_render_ms(PtScene* scene, float* ms);

/* The shape of the handle's most recent wavefront launch (measurement only, not in the reference): persistent
 * blocks of 256 threads per compute unit (= waves per SIMD, of the 8 the hardware holds), compute units of the
 * device, and the bytes of colour workspace the handle holds.  Any pointer may be null. */
int pt_last_launch_shape(PtScene* scene, int32_t* blocks_per_cu, int32_t* compute_units, uint64_t* workspace_bytes);

/* Profiling aid, not in the reference: with PT_STATS=1 in the environment the wavefront kernel
 * counts its phases; `out` holds 16 values: node phases, lanes active in them, leaf phases,
 * lanes, shade phases, lanes, loop iterations, paths started, and the number of boxes the
 * single-precision test skipped although the exact test accepts them (must be 0); the rest
 * are reserved.  All summed over waves. */
int pt_last_render_stats(PtScene* scene, uint64_t* out);

/* Device self-test hook: evaluates one device-side building block of the path
 * on `n` packed input records and copies the results back (host pointers).
 * Not in the reference; it exists so the device arithmetic can be checked
 * against known answers one function at a time. */
enum {
    PT_EVAL_NOISE_1D = 0,          /* in: int32                    out: uint32            (src/rng.cpp:5-20)    */
    PT_EVAL_NOISE_3D = 1,          /* in: int32[3]                 out: uint32            (src/rng.cpp:22-28)   */
    PT_EVAL_XORSHIFT_REAL = 2,     /* in: uint32                   out: uint32,pad,double (src/rng.cpp:63-69, :30-32) */
    PT_EVAL_RANDOM_UNIT_VECTOR = 3,/* in: double[3]                out: double[3]         (src/rng.cpp:34-61)   */
    PT_EVAL_RAY_AABB = 4,          /* in: ray[6], aabb[6]          out: double[2]         (src/geometry.cpp:7-37)   */
    PT_EVAL_RAY_SPHERE = 5,        /* in: ray[6], sphere[4]        out: double[3] (valid,min,max) (src/geometry.cpp:59-82) */
    PT_EVAL_RAY_TRIANGLE = 6,      /* in: ray[6], tri[9]           out: double[2] (valid,t)       (src/geometry.cpp:106-133) */
    PT_EVAL_UNIT_NORMAL = 7,       /* in: tri[9]                   out: double[3]         (src/geometry.cpp:84-90)  */
    PT_EVAL_SCATTER = 8,           /* in: ray[6], material(5 dbl: v[4], type), point[3], normal[3]  out: double[7] (valid, ray[6]) (src/path_tracing.cpp:35-97) */
    PT_EVAL_SQRT_DIV = 9           /* in: double[2] (a,b)          out: double[2] (sqrt(a), a/b) */
};
int pt_device_eval(int32_t device, int32_t op, const void* in, int64_t n, void* out);

#ifdef __cplusplus
}
#endif
#endif /* PATHTRACER_AMD_H */
