// Host entry points kept from the reference: model files, matrices, camera, image output.
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "pt_internal.h"

namespace pt {

static thread_local std::string g_last_error;

int fail(int code, const char* format, ...) {
    char buffer[512];
    va_list args;
    va_start(args, format);
    std::vsnprintf(buffer, sizeof(buffer), format, args);
    va_end(args);
    g_last_error = buffer;
    return code;
}

void clear_error() { g_last_error.clear(); }

PtMat3 multiply(const PtMat3& a, const PtMat3& b) {
    PtMat3 out;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            out.rows[i][j] = a.rows[i][0] * b.rows[0][j] + a.rows[i][1] * b.rows[1][j] + a.rows[i][2] * b.rows[2][j];
    return out;
}

PtMat3 scaling(double x, double y, double z) {
    PtMat3 out{};
    out.rows[0][0] = x;
    out.rows[1][1] = y;
    out.rows[2][2] = z;
    return out;
}

// Rodrigues rotation about the normalised axis, entry for entry as src/linear_algebra.cpp:81-101
// writes it: (1 - cos) * k_i * k_j  +/-  sin * k_l, with cos added on the diagonal.
PtMat3 rotation(double angle, double ax, double ay, double az) {
    const PtVec3 k = normalise(PtVec3{ax, ay, az});
    const double s = std::sin(angle), c = std::cos(angle);
    const double axis[3] = {k.x, k.y, k.z};
    PtMat3 out;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) {
            // symmetric part: the reference multiplies the lower-numbered component first
            const double sym = (1.0f - c) * axis[i < j ? i : j] * axis[i < j ? j : i];
            if (i == j) {
                out.rows[i][j] = sym + c;
            } else {
                const int l = 3 - i - j;                           // the remaining axis
                const bool positive = ((j - i + 3) % 3) == 2;      // (0,2) (1,0) (2,1) carry +sin
                out.rows[i][j] = positive ? sym + s * axis[l] : sym - s * axis[l];
            }
        }
    }
    return out;
}

PtMaterial lambertian(double r, double g, double b) {
    PtMaterial m{};
    m.v[0] = r; m.v[1] = g; m.v[2] = b;
    m.type = PT_LAMBERTIAN;
    return m;
}
PtMaterial metal(double r, double g, double b, double fuzziness) {
    PtMaterial m{};
    m.v[0] = r; m.v[1] = g; m.v[2] = b; m.v[3] = fuzziness;
    m.type = PT_METAL;
    return m;
}
PtMaterial dielectric(double refraction_index) {
    PtMaterial m{};
    m.v[0] = refraction_index;
    m.type = PT_DIELECTRIC;
    return m;
}
PtMaterial diffuse_light(double r, double g, double b, double power) {
    PtMaterial m{};
    m.v[0] = r; m.v[1] = g; m.v[2] = b; m.v[3] = power;
    m.type = PT_DIFFUSE_LIGHT;
    return m;
}

}  // namespace pt

using namespace pt;

extern "C" {

const char* pt_last_error(void) { return g_last_error.c_str(); }

void pt_free(void* p) { std::free(p); }

int pt_load_triangles_file(const char* filename, PtTriangle** triangles, int64_t* count) {
    if (filename == nullptr || triangles == nullptr || count == nullptr) return fail(PT_ERR_INVALID_ARGUMENT, "pt_load_triangles_file: null argument");
    *triangles = nullptr;
    *count = 0;
    FILE* file = std::fopen(filename, "rb");
    if (file == nullptr) return fail(PT_ERR_IO, "cannot open '%s': %s", filename, std::strerror(errno));
    if (std::fseek(file, 0, SEEK_END) != 0) { std::fclose(file); return fail(PT_ERR_IO, "cannot seek '%s'", filename); }
    const long bytes = std::ftell(file);
    std::rewind(file);
    // the reference asserts this (src/model_loading.cpp:26); here it is an error return
    if (bytes < 0 || bytes % static_cast<long>(sizeof(PtTriangle)) != 0) {
        std::fclose(file);
        return fail(PT_ERR_IO, "'%s': %ld bytes is not a whole number of 72-byte triangles", filename, bytes);
    }
    const int64_t n = bytes / static_cast<long>(sizeof(PtTriangle));
    PtTriangle* data = static_cast<PtTriangle*>(std::malloc(n > 0 ? static_cast<size_t>(bytes) : 1));
    if (data == nullptr) { std::fclose(file); return fail(PT_ERR_OUT_OF_MEMORY, "out of memory reading '%s'", filename); }
    const size_t got = std::fread(data, 1, static_cast<size_t>(bytes), file);
    std::fclose(file);
    if (got != static_cast<size_t>(bytes)) { std::free(data); return fail(PT_ERR_IO, "short read on '%s'", filename); }
    *triangles = data;
    *count = n;
    return PT_OK;
}

int pt_save_triangles_file(const char* filename, const PtTriangle* triangles, int64_t count) {
    if (filename == nullptr || (triangles == nullptr && count > 0) || count < 0) return fail(PT_ERR_INVALID_ARGUMENT, "pt_save_triangles_file: bad argument");
    FILE* file = std::fopen(filename, "wb");
    if (file == nullptr) return fail(PT_ERR_IO, "cannot create '%s': %s", filename, std::strerror(errno));
    const size_t wrote = std::fwrite(triangles, sizeof(PtTriangle), static_cast<size_t>(count), file);
    const bool closed = std::fclose(file) == 0;
    if (wrote != static_cast<size_t>(count) || !closed) return fail(PT_ERR_IO, "short write on '%s'", filename);
    return PT_OK;
}

namespace {

// A whitespace-separated word of the file (src/model_loading.cpp:74-89).
struct Word {
    const char* text;
    size_t length;
    // The reference compares only the word's own length (strncmp(word, keyword, word_length) == 0,
    // src/model_loading.cpp:149-163), so a prefix of a keyword counts as the keyword.
    bool is(const char* keyword) const { return std::strncmp(text, keyword, length) == 0; }
};

struct WordReader {
    const std::vector<char>& bytes;
    size_t at = 0;
    static bool blank(char c) { return c == ' ' || c == '\n' || c == '\t' || c == '\r'; }
    Word next() {
        while (at < bytes.size() && blank(bytes[at])) ++at;
        const size_t start = at;
        while (at < bytes.size() && !blank(bytes[at])) ++at;
        return Word{bytes.data() + start, at - start};
    }
    // parse_stl_vec3 (src/model_loading.cpp:91-108): three numbers read with strtod, each of which
    // must use up its whole word
    bool vec3(PtVec3* out) {
        double v[3];
        for (double& x : v) {
            const Word w = next();
            if (w.length == 0) return false;
            char* end = nullptr;
            x = std::strtod(w.text, &end);      // the buffer ends in a NUL, so the last word is terminated too
            if (end != w.text + w.length) return false;
        }
        *out = PtVec3{v[0], v[1], v[2]};
        return true;
    }
};

}  // namespace

int pt_parse_stl_file(const char* filename, PtTriangle** triangles, int64_t* count) {
    if (filename == nullptr || triangles == nullptr || count == nullptr) return fail(PT_ERR_INVALID_ARGUMENT, "pt_parse_stl_file: null argument");
    *triangles = nullptr;
    *count = 0;
    FILE* file = std::fopen(filename, "rb");
    if (file == nullptr) return fail(PT_ERR_IO, "cannot open '%s': %s", filename, std::strerror(errno));
    std::vector<char> bytes;
    char chunk[1 << 16];
    size_t got;
    while ((got = std::fread(chunk, 1, sizeof(chunk), file)) > 0) bytes.insert(bytes.end(), chunk, chunk + got);
    const bool read_error = std::ferror(file) != 0;
    std::fclose(file);
    if (read_error) return fail(PT_ERR_IO, "read error on '%s'", filename);
    bytes.push_back('\0');                      // the reference allocates size + 1 zeroed bytes (:128)
    const size_t text_size = bytes.size() - 1;

    // parse_stl_file (src/model_loading.cpp:138-188): "facet" opens a triangle, "vertex" fills a, b, c in
    // turn, "loop" restarts the vertex count, the normal is read and ignored, other words are skipped.
    // Where the reference asserts, this returns an error.
    std::vector<PtTriangle> found;
    WordReader reader{bytes};
    bool in_facet = false;
    int vertices = 0;
    while (reader.at < text_size) {
        const Word w = reader.next();
        if (w.length == 0) break;
        if (w.is("facet")) {
            if (in_facet) return fail(PT_ERR_IO, "'%s': facet inside a facet (triangle %zu)", filename, found.size());
            in_facet = true;
            found.push_back(PtTriangle{});
        } else if (w.is("endfacet")) {
            if (!in_facet) return fail(PT_ERR_IO, "'%s': endfacet without facet", filename);
            in_facet = false;
        } else if (w.is("normal")) {
            PtVec3 ignored;
            if (!in_facet || !reader.vec3(&ignored)) return fail(PT_ERR_IO, "'%s': bad normal near triangle %zu", filename, found.size());
        } else if (w.is("loop")) {
            vertices = 0;
        } else if (w.is("endloop")) {
            if (vertices != 3) return fail(PT_ERR_IO, "'%s': a loop of %d vertices (triangle %zu)", filename, vertices, found.size());
        } else if (w.is("vertex")) {
            PtVec3 v;
            if (!in_facet || vertices > 2 || !reader.vec3(&v)) return fail(PT_ERR_IO, "'%s': bad vertex near triangle %zu", filename, found.size());
            PtTriangle& t = found.back();
            (vertices == 0 ? t.a : vertices == 1 ? t.b : t.c) = v;
            ++vertices;
        }
    }
    if (!found.empty()) {
        *triangles = static_cast<PtTriangle*>(std::malloc(found.size() * sizeof(PtTriangle)));
        if (*triangles == nullptr) return fail(PT_ERR_IO, "out of memory for %zu triangles", found.size());
        std::memcpy(*triangles, found.data(), found.size() * sizeof(PtTriangle));
    }
    *count = static_cast<int64_t>(found.size());
    return PT_OK;
}

int pt_rotation_matrix(double angle, double axis_x, double axis_y, double axis_z, PtMat3* out) {
    if (out == nullptr) return fail(PT_ERR_INVALID_ARGUMENT, "pt_rotation_matrix: null output");
    *out = rotation(angle, axis_x, axis_y, axis_z);
    return PT_OK;
}

int pt_look_at_matrix(const PtVec3* position, const PtVec3* target, PtMat3* out) {
    if (position == nullptr || target == nullptr || out == nullptr) return fail(PT_ERR_INVALID_ARGUMENT, "pt_look_at_matrix: null argument");
    // src/linear_algebra.cpp:103-122: the matrix columns are the camera's x, y and z axes
    const PtVec3 z_axis = normalise(*position - *target);
    const PtVec3 x_axis = normalise(cross(PtVec3{0.0f, 1.0f, 0.0f}, z_axis));
    const PtVec3 y_axis = cross(z_axis, x_axis);
    const PtVec3 axes[3] = {x_axis, y_axis, z_axis};
    for (int column = 0; column < 3; ++column) {
        out->rows[0][column] = axes[column].x;
        out->rows[1][column] = axes[column].y;
        out->rows[2][column] = axes[column].z;
    }
    return PT_OK;
}

int pt_camera_look_at(const PtVec3* position, const PtVec3* target, double fov_y, double aperture,
                      double focus_distance, PtCamera* out) {
    if (position == nullptr || target == nullptr || out == nullptr) return fail(PT_ERR_INVALID_ARGUMENT, "pt_camera_look_at: null argument");
    std::memset(out, 0, sizeof(*out));
    pt_look_at_matrix(position, target, &out->orientation);
    out->target = *target;
    out->distance = magnitude(*position - *target);
    out->fov_y = fov_y;
    out->aperture = aperture;
    out->focus_distance = (focus_distance > 0.0) ? focus_distance : out->distance;
    return PT_OK;
}

int pt_camera_frame(const PtCamera* camera, double aspect_ratio, int32_t width, int32_t height, PtFrame* frame) {
    if (camera == nullptr || frame == nullptr) return fail(PT_ERR_INVALID_ARGUMENT, "pt_camera_frame: null argument");
    if (width < 2 || height < 2) return fail(PT_ERR_INVALID_ARGUMENT, "pt_camera_frame: width and height must be at least 2 (u = column / (width - 1))");
    const PtMat3& o = camera->orientation;
    const PtVec3 camera_x{o.rows[0][0], o.rows[1][0], o.rows[2][0]};   // get_column, src/main.cpp:928-930
    const PtVec3 camera_y{o.rows[0][1], o.rows[1][1], o.rows[2][1]};
    const PtVec3 camera_z{o.rows[0][2], o.rows[1][2], o.rows[2][2]};
    // get_position, src/path_tracing.cpp:229-231
    const PtVec3 position = transform(o, PtVec3{0.0f, 0.0f, camera->distance}) + camera->target;
    // src/main.cpp:989-994
    const double viewport_height = 2.0f * std::tan(0.5f * camera->fov_y);
    const double viewport_width = aspect_ratio * viewport_height;
    const PtVec3 step_x = (camera->focus_distance * viewport_width) * camera_x;
    const PtVec3 step_y = (camera->focus_distance * viewport_height) * camera_y;
    const PtVec3 bottom_left = position - 0.5f * step_x - 0.5f * step_y - camera->focus_distance * camera_z;

    std::memset(frame, 0, sizeof(*frame));
    frame->width = width;
    frame->height = height;
    frame->row_begin = 0;
    frame->row_end = height;
    frame->sample_begin = 0;
    frame->sample_count = 1;
    frame->shard_index = 0; frame->shard_count = 1;
    frame->max_bounces = 50;            // MAX_BOUNCE_COUNT, src/path_tracing.cpp:172
    frame->aperture = camera->aperture;
    frame->camera_position = position;
    frame->camera_x = camera_x;
    frame->camera_y = camera_y;
    frame->bottom_left = bottom_left;
    frame->step_x = step_x;
    frame->step_y = step_y;
    return PT_OK;
}

int pt_camera_apply_input(PtCamera* camera, const PtViewerInput* in, int32_t* camera_modified) {
    if (camera == nullptr || in == nullptr || camera_modified == nullptr) return fail(PT_ERR_INVALID_ARGUMENT, "pt_camera_apply_input: null argument");
    bool modified = false;
    const double dt = in->frame_duration;
    if (in->left_button_held != 0 && (in->mouse_delta_x != 0 || in->mouse_delta_y != 0)) {            // src/main.cpp:898-908
        const double sensitivity = 4.0f;
        const double yaw = static_cast<double>(in->mouse_delta_x) * sensitivity * dt;
        const double pitch = -static_cast<double>(in->mouse_delta_y) * sensitivity * dt;
        const PtMat3 yaw_rotation = rotation(degrees_to_radians(yaw), 0.0f, 0.0f, 1.0f);
        const PtMat3 pitch_rotation = rotation(degrees_to_radians(pitch), 1.0f, 0.0f, 0.0f);
        camera->orientation = multiply(multiply(yaw_rotation, camera->orientation), pitch_rotation);
        modified = true;
    }
    const float scroll_sensitivity = 0.95f;                                                           // :910-924
    for (int i = 0; i < in->scroll; ++i) camera->distance *= scroll_sensitivity;
    for (int i = in->scroll; i < 0; ++i) camera->distance /= scroll_sensitivity;
    modified = modified || in->scroll != 0;
    const PtMat3& o = camera->orientation;                                                            // :928-929, after the drag
    const PtVec3 camera_x{o.rows[0][0], o.rows[1][0], o.rows[2][0]};
    const PtVec3 camera_y{o.rows[0][1], o.rows[1][1], o.rows[2][1]};
    const float camera_speed = 0.6f;                                                                  // :932-952
    if (in->key_a != 0) { camera->target = camera->target - (camera->distance * camera_speed * dt) * camera_x; modified = true; }
    if (in->key_d != 0) { camera->target = camera->target + (camera->distance * camera_speed * dt) * camera_x; modified = true; }
    if (in->key_w != 0) { camera->target = camera->target + (camera->distance * camera_speed * dt) * camera_y; modified = true; }
    if (in->key_s != 0) { camera->target = camera->target - (camera->distance * camera_speed * dt) * camera_y; modified = true; }
    const double fov_sensitivity = 0.1f, focus_sensitivity = 1.0f, aperture_sensitivity = 0.1f;       // :954-985
    if (in->key_q != 0) { camera->fov_y -= fov_sensitivity * dt; modified = true; }
    if (in->key_e != 0) { camera->fov_y += fov_sensitivity * dt; modified = true; }
    if (in->key_up != 0) { camera->focus_distance += focus_sensitivity * dt * camera->distance; modified = true; }
    if (in->key_down != 0) { camera->focus_distance -= focus_sensitivity * dt * camera->distance; modified = true; }
    if (in->key_left != 0) { camera->aperture -= aperture_sensitivity * dt; modified = true; }
    if (in->key_right != 0) { camera->aperture += aperture_sensitivity * dt; modified = true; }
    *camera_modified = modified ? 1 : 0;
    return PT_OK;
}

int pt_tonemap_bgra8(const double* accum, int64_t pixel_count, int32_t sample_count, uint8_t* out) {
    if (accum == nullptr || out == nullptr || pixel_count < 0 || sample_count <= 0) return fail(PT_ERR_INVALID_ARGUMENT, "pt_tonemap_bgra8: bad argument");
    const double samples = static_cast<double>(sample_count);
    for (int64_t i = 0; i < pixel_count; ++i) {
        for (int channel = 0; channel < 3; ++channel) {
            // src/main.cpp:1026-1032: std::min<real>(sqrt(sum / n), 1.0f), then 255.0f * v truncated
            const double level = min_of(std::sqrt(accum[4 * i + channel] / samples), 1.0f);
            out[4 * i + channel] = static_cast<unsigned char>(255.0f * level);
        }
        out[4 * i + 3] = 255;
    }
    return PT_OK;
}

int pt_write_pixel_data_file(const char* filename, const uint8_t* bgra, int64_t pixel_count) {
    if (filename == nullptr || bgra == nullptr || pixel_count < 0) return fail(PT_ERR_INVALID_ARGUMENT, "pt_write_pixel_data_file: bad argument");
    // src/main.cpp:192-197 swaps B and R in place, writes, and swaps back; writing a swapped copy
    // leaves the same bytes on disk.
    std::vector<uint8_t> rgba(bgra, bgra + 4 * pixel_count);
    for (int64_t i = 0; i < pixel_count; ++i) {
        rgba[4 * i + 0] = bgra[4 * i + 2];
        rgba[4 * i + 2] = bgra[4 * i + 0];
    }
    FILE* file = std::fopen(filename, "wb");
    if (file == nullptr) return fail(PT_ERR_IO, "cannot create '%s': %s", filename, std::strerror(errno));
    const size_t wrote = std::fwrite(rgba.data(), 1, rgba.size(), file);
    const bool closed = std::fclose(file) == 0;
    if (wrote != rgba.size() || !closed) return fail(PT_ERR_IO, "short write on '%s'", filename);
    return PT_OK;
}

int pt_write_ppm_file(const char* filename, const uint8_t* bgra, int32_t width, int32_t height) {
    if (filename == nullptr || bgra == nullptr || width <= 0 || height <= 0) return fail(PT_ERR_INVALID_ARGUMENT, "pt_write_ppm_file: bad argument");
    // the buffer's row 0 is the bottom of the picture (src/main.cpp:90-97: v grows with the row and the
    // camera's y axis); PPM wants the top row first
    std::vector<uint8_t> rgb(3ull * width * height);
    for (int32_t row = 0; row < height; ++row) {
        const uint8_t* in = bgra + 4ull * (height - 1 - row) * width;
        uint8_t* out = rgb.data() + 3ull * row * width;
        for (int32_t column = 0; column < width; ++column) {
            out[3 * column + 0] = in[4 * column + 2];
            out[3 * column + 1] = in[4 * column + 1];
            out[3 * column + 2] = in[4 * column + 0];
        }
    }
    FILE* file = std::fopen(filename, "wb");
    if (file == nullptr) return fail(PT_ERR_IO, "cannot create '%s': %s", filename, std::strerror(errno));
    const bool header = std::fprintf(file, "P6\n%d %d\n255\n", width, height) > 0;
    const size_t wrote = std::fwrite(rgb.data(), 1, rgb.size(), file);
    const bool closed = std::fclose(file) == 0;
    if (!header || wrote != rgb.size() || !closed) return fail(PT_ERR_IO, "short write on '%s'", filename);
    return PT_OK;
}

}  // extern "C"
