// Node-API addon: the JavaScript host's door to the C ABI of include/flowmpm.h.
//
// The reference's simulator hands its kernels to three.js
// (src/mls-mpm/mlsMpmSimulator.js:2538-2539); host_js/simulator.js calls these
// functions instead.  The addon is a thin marshalling layer: typed arrays in,
// status codes turned into JavaScript exceptions, no computation.  It links
// libflowmpm.so (flow_amd/lib) through an $ORIGIN-relative rpath.
#include <node_api.h>

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "flowmpm.h"

namespace {

#define NAPI_OK(call)                                                      \
    do {                                                                   \
        if ((call) != napi_ok) {                                           \
            napi_throw_error(env, nullptr, "N-API call failed: " #call);   \
            return nullptr;                                                \
        }                                                                  \
    } while (0)

napi_value throw_flowmpm(napi_env env, flowmpm_sim *sim, int code)
{
    std::string msg = "flowmpm error " + std::to_string(code) + ": " + flowmpm_last_error(sim);
    napi_throw_error(env, std::to_string(code).c_str(), msg.c_str());
    return nullptr;
}

void finalize_handle(napi_env, void *data, void *) { flowmpm_destroy(static_cast<flowmpm_sim *>(data)); }

bool get_args(napi_env env, napi_callback_info info, size_t want, napi_value *argv)
{
    size_t argc = want;
    if (napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr) != napi_ok || argc < want) {
        napi_throw_type_error(env, nullptr, "wrong number of arguments");
        return false;
    }
    return true;
}

flowmpm_sim *unwrap(napi_env env, napi_value v)
{
    void *p = nullptr;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p) {
        napi_throw_type_error(env, nullptr, "expected a flowmpm handle");
        return nullptr;
    }
    return static_cast<flowmpm_sim *>(p);
}

bool get_number(napi_env env, napi_value obj, const char *key, double *out, bool required = false)
{
    bool has = false;
    napi_value v;
    if (napi_has_named_property(env, obj, key, &has) != napi_ok || !has) {
        if (required) napi_throw_type_error(env, nullptr, (std::string("missing property ") + key).c_str());
        return false;
    }
    napi_get_named_property(env, obj, key, &v);
    return napi_get_value_double(env, v, out) == napi_ok;
}

void get_vec3(napi_env env, napi_value obj, const char *key, float out[3])
{
    bool has = false;
    if (napi_has_named_property(env, obj, key, &has) != napi_ok || !has) return;
    napi_value arr, e;
    napi_get_named_property(env, obj, key, &arr);
    for (uint32_t i = 0; i < 3; i++) {
        double d = 0;
        if (napi_get_element(env, arr, i, &e) == napi_ok && napi_get_value_double(env, e, &d) == napi_ok)
            out[i] = (float)d;
    }
}

// Typed array -> pointer + element count; `type` must match.
bool typed(napi_env env, napi_value v, napi_typedarray_type type, void **data, size_t *len)
{
    napi_typedarray_type t;
    napi_value buf;
    size_t off;
    if (napi_get_typedarray_info(env, v, &t, len, data, &buf, &off) != napi_ok || t != type) {
        napi_throw_type_error(env, nullptr, "expected a typed array of the right element type");
        return false;
    }
    return true;
}

napi_value Create(napi_env env, napi_callback_info info)
{
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return nullptr;
    flowmpm_config cfg;
    std::memset(&cfg, 0, sizeof cfg);
    double d;
    if (!get_number(env, argv[0], "gridX", &d, true)) return nullptr;
    cfg.grid_x = (int32_t)d;
    if (!get_number(env, argv[0], "gridY", &d, true)) return nullptr;
    cfg.grid_y = (int32_t)d;
    if (!get_number(env, argv[0], "gridZ", &d, true)) return nullptr;
    cfg.grid_z = (int32_t)d;
    if (!get_number(env, argv[0], "maxParticles", &d, true)) return nullptr;
    cfg.max_particles = (int64_t)d;
    cfg.algorithm = FLOWMPM_ALGO_TILED;
    cfg.rebin_interval = 1;
    if (get_number(env, argv[0], "device", &d)) cfg.device = (int32_t)d;
    if (get_number(env, argv[0], "algorithm", &d)) cfg.algorithm = (int32_t)d;
    if (get_number(env, argv[0], "rebinInterval", &d)) cfg.rebin_interval = (int32_t)d;
    flowmpm_sim *sim = nullptr;
    int rc = flowmpm_create(&cfg, &sim);
    if (rc != FLOWMPM_OK) return throw_flowmpm(env, nullptr, rc);
    napi_value out;
    NAPI_OK(napi_create_external(env, sim, finalize_handle, nullptr, &out));
    return out;
}

napi_value UploadParticles(napi_env env, napi_callback_info info)
{
    napi_value argv[3];
    if (!get_args(env, info, 3, argv)) return nullptr;
    flowmpm_sim *sim = unwrap(env, argv[0]);
    if (!sim) return nullptr;
    void *data;
    size_t len;
    if (!typed(env, argv[1], napi_float32_array, &data, &len)) return nullptr;
    int64_t count;
    NAPI_OK(napi_get_value_int64(env, argv[2], &count));
    if (count < 0 || (size_t)count * FLOWMPM_PARTICLE_STRIDE > len) {
        napi_throw_range_error(env, nullptr, "particle count exceeds the array");
        return nullptr;
    }
    int rc = flowmpm_upload_particles(sim, static_cast<const float *>(data), nullptr, count);
    if (rc != FLOWMPM_OK) return throw_flowmpm(env, sim, rc);
    return nullptr;
}

napi_value SetNumParticles(napi_env env, napi_callback_info info)
{
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return nullptr;
    flowmpm_sim *sim = unwrap(env, argv[0]);
    if (!sim) return nullptr;
    int64_t count;
    NAPI_OK(napi_get_value_int64(env, argv[1], &count));
    int rc = flowmpm_set_num_particles(sim, count);
    if (rc != FLOWMPM_OK) return throw_flowmpm(env, sim, rc);
    return nullptr;
}

void read_uniforms(napi_env env, napi_value obj, flowmpm_uniforms *out)
{
    flowmpm_uniforms u;
    std::memset(&u, 0, sizeof u);
    double d;
    if (get_number(env, obj, "dt", &d)) u.dt = (float)d;
    if (get_number(env, obj, "stiffness", &d)) u.stiffness = (float)d;
    if (get_number(env, obj, "restDensity", &d)) u.rest_density = (float)d;
    if (get_number(env, obj, "dynamicViscosity", &d)) u.dynamic_viscosity = (float)d;
    if (get_number(env, obj, "noise", &d)) u.noise = (float)d;
    if (get_number(env, obj, "time", &d)) u.time = (float)d;
    if (get_number(env, obj, "cursorInteraction", &d)) u.cursor_interaction = (uint32_t)d;
    if (get_number(env, obj, "hidePercentage", &d)) u.hide_percentage = (float)d;
    if (get_number(env, obj, "gridMode", &d)) u.grid_mode = (uint32_t)d;
    if (get_number(env, obj, "frontGravityMode", &d)) u.front_gravity_mode = (uint32_t)d;
    get_vec3(env, obj, "mouseRayDirection", u.mouse_ray_direction);
    get_vec3(env, obj, "mouseRayOrigin", u.mouse_ray_origin);
    get_vec3(env, obj, "mouseForce", u.mouse_force);
    *out = u;
}

napi_value SetUniforms(napi_env env, napi_callback_info info)
{
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return nullptr;
    flowmpm_sim *sim = unwrap(env, argv[0]);
    if (!sim) return nullptr;
    flowmpm_uniforms u;
    read_uniforms(env, argv[1], &u);
    int rc = flowmpm_set_uniforms(sim, &u);
    if (rc != FLOWMPM_OK) return throw_flowmpm(env, sim, rc);
    return nullptr;
}

// setModeUniforms(handle, ArrayBuffer): the buffer holds one flowmpm_mode_uniforms record
// (host_js/modes.js packs it; every field is a 4-byte word, so there is no padding).
napi_value SetModeUniforms(napi_env env, napi_callback_info info)
{
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return nullptr;
    flowmpm_sim *sim = unwrap(env, argv[0]);
    if (!sim) return nullptr;
    void *data = nullptr;
    size_t bytes = 0;
    if (napi_get_arraybuffer_info(env, argv[1], &data, &bytes) != napi_ok || bytes != sizeof(flowmpm_mode_uniforms)) {
        napi_throw_type_error(env, nullptr, "expected an ArrayBuffer of sizeof(flowmpm_mode_uniforms) bytes");
        return nullptr;
    }
    flowmpm_mode_uniforms m;
    std::memcpy(&m, data, sizeof m);
    int rc = flowmpm_set_mode_uniforms(sim, &m);
    if (rc != FLOWMPM_OK) return throw_flowmpm(env, sim, rc);
    return nullptr;
}

napi_value ModeUniformsSize(napi_env env, napi_callback_info)
{
    napi_value out;
    NAPI_OK(napi_create_int32(env, (int32_t)sizeof(flowmpm_mode_uniforms), &out));
    return out;
}

napi_value Step(napi_env env, napi_callback_info info)
{
    napi_value argv[3];
    if (!get_args(env, info, 3, argv)) return nullptr;
    flowmpm_sim *sim = unwrap(env, argv[0]);
    if (!sim) return nullptr;
    int32_t n;
    double dtime;
    NAPI_OK(napi_get_value_int32(env, argv[1], &n));
    NAPI_OK(napi_get_value_double(env, argv[2], &dtime));
    int rc = flowmpm_step(sim, n, dtime);
    if (rc != FLOWMPM_OK) return throw_flowmpm(env, sim, rc);
    return nullptr;
}

napi_value Synchronize(napi_env env, napi_callback_info info)
{
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return nullptr;
    flowmpm_sim *sim = unwrap(env, argv[0]);
    if (!sim) return nullptr;
    int rc = flowmpm_synchronize(sim);
    if (rc != FLOWMPM_OK) return throw_flowmpm(env, sim, rc);
    return nullptr;
}

napi_value DownloadParticles(napi_env env, napi_callback_info info)
{
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return nullptr;
    flowmpm_sim *sim = unwrap(env, argv[0]);
    if (!sim) return nullptr;
    void *data;
    size_t len;
    if (!typed(env, argv[1], napi_float32_array, &data, &len)) return nullptr;
    int64_t count = 0;
    int rc = flowmpm_download_particles(sim, static_cast<float *>(data), nullptr,
                                        (int64_t)(len / FLOWMPM_PARTICLE_STRIDE), &count);
    if (rc != FLOWMPM_OK) return throw_flowmpm(env, sim, rc);
    napi_value out;
    NAPI_OK(napi_create_int64(env, count, &out));
    return out;
}

napi_value DownloadField(napi_env env, napi_callback_info info)
{
    napi_value argv[3];
    if (!get_args(env, info, 3, argv)) return nullptr;
    flowmpm_sim *sim = unwrap(env, argv[0]);
    if (!sim) return nullptr;
    int32_t field;
    NAPI_OK(napi_get_value_int32(env, argv[1], &field));
    void *data;
    size_t len;
    if (!typed(env, argv[2], napi_float32_array, &data, &len)) return nullptr;
    const int width = (field == FLOWMPM_FIELD_DENSITY || field == FLOWMPM_FIELD_MASS) ? 1 : 3;
    int rc = flowmpm_download_field(sim, field, static_cast<float *>(data), (int64_t)(len / width));
    if (rc != FLOWMPM_OK) return throw_flowmpm(env, sim, rc);
    return nullptr;
}

napi_value DownloadGrid(napi_env env, napi_callback_info info, bool integer)
{
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return nullptr;
    flowmpm_sim *sim = unwrap(env, argv[0]);
    if (!sim) return nullptr;
    void *data;
    size_t len;
    if (!typed(env, argv[1], integer ? napi_int32_array : napi_float32_array, &data, &len)) return nullptr;
    int rc = integer ? flowmpm_download_grid_i32(sim, static_cast<int32_t *>(data), (int64_t)(len / 4))
                     : flowmpm_download_grid_f32(sim, static_cast<float *>(data), (int64_t)(len / 4));
    if (rc != FLOWMPM_OK) return throw_flowmpm(env, sim, rc);
    return nullptr;
}

napi_value DownloadGridI32(napi_env env, napi_callback_info info) { return DownloadGrid(env, info, true); }
napi_value DownloadGridF32(napi_env env, napi_callback_info info) { return DownloadGrid(env, info, false); }

napi_value DeviceCount(napi_env env, napi_callback_info)
{
    int n = 0;
    flowmpm_device_count(&n);     // 0 devices is an answer here, not an exception
    napi_value out;
    NAPI_OK(napi_create_int32(env, n, &out));
    return out;
}

napi_value Version(napi_env env, napi_callback_info)
{
    napi_value out;
    NAPI_OK(napi_create_int32(env, flowmpm_version(), &out));
    return out;
}

// ---- slab groups: a decomposed domain behind one object (include/flowmpm.h, "slab groups") ----
// The external value of a group is tagged by a wrapper struct so that a handle cannot be passed for it.
struct GroupBox { flowmpm_slab_group *group; };

void finalize_group(napi_env, void *data, void *)
{
    GroupBox *box = static_cast<GroupBox *>(data);
    flowmpm_slab_group_destroy(box->group);
    delete box;
}

flowmpm_slab_group *unwrap_group(napi_env env, napi_value v)
{
    void *p = nullptr;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p) {
        napi_throw_type_error(env, nullptr, "expected a flowmpm slab group");
        return nullptr;
    }
    return static_cast<GroupBox *>(p)->group;
}

napi_value throw_group(napi_env env, flowmpm_slab_group *g, int code)
{
    std::string msg = "flowmpm error " + std::to_string(code) + ": " + flowmpm_slab_group_last_error(g);
    napi_throw_error(env, std::to_string(code).c_str(), msg.c_str());
    return nullptr;
}

bool get_int_list(napi_env env, napi_value obj, const char *key, std::vector<int64_t> *out)
{
    bool has = false;
    if (napi_has_named_property(env, obj, key, &has) != napi_ok || !has) return false;
    napi_value arr, e;
    uint32_t n = 0;
    napi_get_named_property(env, obj, key, &arr);
    if (napi_get_array_length(env, arr, &n) != napi_ok) return false;
    for (uint32_t i = 0; i < n; i++) {
        double d = 0;
        if (napi_get_element(env, arr, i, &e) != napi_ok || napi_get_value_double(env, e, &d) != napi_ok) return false;
        out->push_back((int64_t)d);
    }
    return true;
}

// groupCreate({gridX, gridY, gridZ, planeStarts: [...], maxParticles: [...per slab], devices?: [...], rebinInterval?})
napi_value GroupCreate(napi_env env, napi_callback_info info)
{
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return nullptr;
    flowmpm_config cfg;
    std::memset(&cfg, 0, sizeof cfg);
    double d;
    if (!get_number(env, argv[0], "gridX", &d, true)) return nullptr;
    cfg.grid_x = (int32_t)d;
    if (!get_number(env, argv[0], "gridY", &d, true)) return nullptr;
    cfg.grid_y = (int32_t)d;
    if (!get_number(env, argv[0], "gridZ", &d, true)) return nullptr;
    cfg.grid_z = (int32_t)d;
    cfg.algorithm = FLOWMPM_ALGO_TILED;
    cfg.rebin_interval = 64;
    if (get_number(env, argv[0], "rebinInterval", &d)) cfg.rebin_interval = (int32_t)d;
    std::vector<int64_t> starts, caps, devs;
    if (!get_int_list(env, argv[0], "planeStarts", &starts) || starts.size() < 2 ||
        !get_int_list(env, argv[0], "maxParticles", &caps) || caps.size() != starts.size() - 1) {
        napi_throw_type_error(env, nullptr, "planeStarts needs n + 1 entries and maxParticles n");
        return nullptr;
    }
    const bool placed = get_int_list(env, argv[0], "devices", &devs);
    if (placed && devs.size() != caps.size()) {
        napi_throw_type_error(env, nullptr, "devices needs one entry per slab");
        return nullptr;
    }
    std::vector<int32_t> starts32(starts.begin(), starts.end()), devs32(devs.begin(), devs.end());
    flowmpm_slab_group *g = nullptr;
    int rc = flowmpm_slab_group_create(&cfg, starts32.data(), (int32_t)caps.size(), placed ? devs32.data() : nullptr, caps.data(), &g);
    if (rc != FLOWMPM_OK) return throw_group(env, nullptr, rc);
    napi_value out;
    GroupBox *box = new GroupBox{g};
    if (napi_create_external(env, box, finalize_group, nullptr, &out) != napi_ok) {
        finalize_group(env, box, nullptr);
        napi_throw_error(env, nullptr, "napi_create_external failed");
        return nullptr;
    }
    return out;
}

napi_value GroupUpload(napi_env env, napi_callback_info info)
{
    napi_value argv[3];
    if (!get_args(env, info, 3, argv)) return nullptr;
    flowmpm_slab_group *g = unwrap_group(env, argv[0]);
    if (!g) return nullptr;
    void *data;
    size_t len;
    if (!typed(env, argv[1], napi_float32_array, &data, &len)) return nullptr;
    int64_t count;
    NAPI_OK(napi_get_value_int64(env, argv[2], &count));
    if (count < 0 || (size_t)count * FLOWMPM_PARTICLE_STRIDE > len) {
        napi_throw_range_error(env, nullptr, "particle count exceeds the array");
        return nullptr;
    }
    int rc = flowmpm_slab_group_upload(g, static_cast<const float *>(data), count);
    if (rc != FLOWMPM_OK) return throw_group(env, g, rc);
    return nullptr;
}

napi_value GroupGenerate(napi_env env, napi_callback_info info)
{
    napi_value argv[5];
    if (!get_args(env, info, 5, argv)) return nullptr;
    flowmpm_slab_group *g = unwrap_group(env, argv[0]);
    if (!g) return nullptr;
    int32_t scene;
    uint32_t seed;
    int64_t count;
    double fill;
    NAPI_OK(napi_get_value_int32(env, argv[1], &scene));
    NAPI_OK(napi_get_value_uint32(env, argv[2], &seed));
    NAPI_OK(napi_get_value_int64(env, argv[3], &count));
    NAPI_OK(napi_get_value_double(env, argv[4], &fill));
    int rc = flowmpm_slab_group_generate(g, scene, seed, count, fill);
    if (rc != FLOWMPM_OK) return throw_group(env, g, rc);
    return nullptr;
}

napi_value GroupSetUniforms(napi_env env, napi_callback_info info)
{
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return nullptr;
    flowmpm_slab_group *g = unwrap_group(env, argv[0]);
    if (!g) return nullptr;
    flowmpm_uniforms u;
    read_uniforms(env, argv[1], &u);
    int rc = flowmpm_slab_group_set_uniforms(g, &u);
    if (rc != FLOWMPM_OK) return throw_group(env, g, rc);
    return nullptr;
}

napi_value GroupSetModeUniforms(napi_env env, napi_callback_info info)
{
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return nullptr;
    flowmpm_slab_group *g = unwrap_group(env, argv[0]);
    if (!g) return nullptr;
    void *data = nullptr;
    size_t bytes = 0;
    if (napi_get_arraybuffer_info(env, argv[1], &data, &bytes) != napi_ok || bytes != sizeof(flowmpm_mode_uniforms)) {
        napi_throw_type_error(env, nullptr, "expected an ArrayBuffer of sizeof(flowmpm_mode_uniforms) bytes");
        return nullptr;
    }
    flowmpm_mode_uniforms m;
    std::memcpy(&m, data, sizeof m);
    int rc = flowmpm_slab_group_set_mode_uniforms(g, &m);
    if (rc != FLOWMPM_OK) return throw_group(env, g, rc);
    return nullptr;
}

napi_value GroupStep(napi_env env, napi_callback_info info)
{
    napi_value argv[3];
    if (!get_args(env, info, 3, argv)) return nullptr;
    flowmpm_slab_group *g = unwrap_group(env, argv[0]);
    if (!g) return nullptr;
    int32_t n;
    double dtime;
    NAPI_OK(napi_get_value_int32(env, argv[1], &n));
    NAPI_OK(napi_get_value_double(env, argv[2], &dtime));
    int rc = flowmpm_slab_group_step(g, n, dtime);
    if (rc != FLOWMPM_OK) return throw_group(env, g, rc);
    return nullptr;
}

napi_value GroupSynchronize(napi_env env, napi_callback_info info)
{
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return nullptr;
    flowmpm_slab_group *g = unwrap_group(env, argv[0]);
    if (!g) return nullptr;
    int rc = flowmpm_slab_group_synchronize(g);
    if (rc != FLOWMPM_OK) return throw_group(env, g, rc);
    return nullptr;
}

napi_value GroupDownloadParticles(napi_env env, napi_callback_info info)
{
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return nullptr;
    flowmpm_slab_group *g = unwrap_group(env, argv[0]);
    if (!g) return nullptr;
    void *data;
    size_t len;
    if (!typed(env, argv[1], napi_float32_array, &data, &len)) return nullptr;
    int64_t count = 0;
    int rc = flowmpm_slab_group_download_particles(g, static_cast<float *>(data), (int64_t)(len / FLOWMPM_PARTICLE_STRIDE), &count);
    if (rc != FLOWMPM_OK) return throw_group(env, g, rc);
    napi_value out;
    NAPI_OK(napi_create_int64(env, count, &out));
    return out;
}

napi_value GroupDownloadField(napi_env env, napi_callback_info info)
{
    napi_value argv[3];
    if (!get_args(env, info, 3, argv)) return nullptr;
    flowmpm_slab_group *g = unwrap_group(env, argv[0]);
    if (!g) return nullptr;
    int32_t field;
    NAPI_OK(napi_get_value_int32(env, argv[1], &field));
    void *data;
    size_t len;
    if (!typed(env, argv[2], napi_float32_array, &data, &len)) return nullptr;
    const int width = (field == FLOWMPM_FIELD_DENSITY || field == FLOWMPM_FIELD_MASS) ? 1 : 3;
    int rc = flowmpm_slab_group_download_field(g, field, static_cast<float *>(data), (int64_t)(len / width));
    if (rc != FLOWMPM_OK) return throw_group(env, g, rc);
    return nullptr;
}

// groupInfo(group) -> {slabs, rccl, particles, perSlab: [...], conservation: [x, y, z, mass, particles], migrated, maxHop}
// (64-bit sums as doubles: exact below 2^53, which 2^27 particles x 1e7 stay under)
napi_value GroupInfo(napi_env env, napi_callback_info info)
{
    napi_value argv[1];
    if (!get_args(env, info, 1, argv)) return nullptr;
    flowmpm_slab_group *g = unwrap_group(env, argv[0]);
    if (!g) return nullptr;
    int32_t n = 0, rccl = 0, hop = 0;
    int64_t total = 0, migrated = 0, sums[5];
    int rc = flowmpm_slab_group_size(g, &n, &rccl);
    std::vector<int64_t> per((size_t)(n > 0 ? n : 1));
    if (rc == FLOWMPM_OK) rc = flowmpm_slab_group_get_num_particles(g, &total, per.data());
    if (rc == FLOWMPM_OK) rc = flowmpm_slab_group_conservation(g, sums);
    if (rc == FLOWMPM_OK) rc = flowmpm_slab_group_migration_stats(g, &migrated, &hop);
    if (rc != FLOWMPM_OK) return throw_group(env, g, rc);
    napi_value out, v, arr;
    NAPI_OK(napi_create_object(env, &out));
    NAPI_OK(napi_create_int32(env, n, &v));
    NAPI_OK(napi_set_named_property(env, out, "slabs", v));
    NAPI_OK(napi_get_boolean(env, rccl != 0, &v));
    NAPI_OK(napi_set_named_property(env, out, "rccl", v));
    NAPI_OK(napi_create_double(env, (double)total, &v));
    NAPI_OK(napi_set_named_property(env, out, "particles", v));
    NAPI_OK(napi_create_array_with_length(env, (size_t)n, &arr));
    for (int32_t r = 0; r < n; r++) {
        NAPI_OK(napi_create_double(env, (double)per[(size_t)r], &v));
        NAPI_OK(napi_set_element(env, arr, (uint32_t)r, v));
    }
    NAPI_OK(napi_set_named_property(env, out, "perSlab", arr));
    NAPI_OK(napi_create_array_with_length(env, 5, &arr));
    for (uint32_t k = 0; k < 5; k++) {
        NAPI_OK(napi_create_double(env, (double)sums[k], &v));
        NAPI_OK(napi_set_element(env, arr, k, v));
    }
    NAPI_OK(napi_set_named_property(env, out, "conservation", arr));
    NAPI_OK(napi_create_double(env, (double)migrated, &v));
    NAPI_OK(napi_set_named_property(env, out, "migrated", v));
    NAPI_OK(napi_create_int32(env, hop, &v));
    NAPI_OK(napi_set_named_property(env, out, "maxHop", v));
    return out;
}

// groupSlabGridI32(group, r, Int32Array): the local planes of slab r as {x, y, z, mass} records
napi_value GroupSlabGridI32(napi_env env, napi_callback_info info)
{
    napi_value argv[3];
    if (!get_args(env, info, 3, argv)) return nullptr;
    flowmpm_slab_group *g = unwrap_group(env, argv[0]);
    if (!g) return nullptr;
    int32_t r;
    NAPI_OK(napi_get_value_int32(env, argv[1], &r));
    void *data;
    size_t len;
    if (!typed(env, argv[2], napi_int32_array, &data, &len)) return nullptr;
    flowmpm_sim *sim = nullptr;
    int rc = flowmpm_slab_group_slab(g, r, &sim);
    if (rc != FLOWMPM_OK) return throw_group(env, g, rc);
    if ((rc = flowmpm_slab_group_synchronize(g)) != FLOWMPM_OK) return throw_group(env, g, rc);
    rc = flowmpm_download_grid_i32(sim, static_cast<int32_t *>(data), (int64_t)(len / 4));
    if (rc != FLOWMPM_OK) return throw_flowmpm(env, sim, rc);
    return nullptr;
}

napi_value RcclSelftest(napi_env env, napi_callback_info info)
{
    napi_value argv[2];
    if (!get_args(env, info, 2, argv)) return nullptr;
    int32_t device;
    int64_t words;
    NAPI_OK(napi_get_value_int32(env, argv[0], &device));
    NAPI_OK(napi_get_value_int64(env, argv[1], &words));
    int rc = flowmpm_rccl_selftest(device, words);
    if (rc != FLOWMPM_OK) return throw_flowmpm(env, nullptr, rc);
    return nullptr;
}

napi_value Init(napi_env env, napi_value exports)
{
    const napi_property_descriptor props[] = {
        {"create", nullptr, Create, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"uploadParticles", nullptr, UploadParticles, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"setNumParticles", nullptr, SetNumParticles, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"setUniforms", nullptr, SetUniforms, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"setModeUniforms", nullptr, SetModeUniforms, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"modeUniformsSize", nullptr, ModeUniformsSize, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"step", nullptr, Step, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"synchronize", nullptr, Synchronize, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"downloadParticles", nullptr, DownloadParticles, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"downloadField", nullptr, DownloadField, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"downloadGridI32", nullptr, DownloadGridI32, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"downloadGridF32", nullptr, DownloadGridF32, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"deviceCount", nullptr, DeviceCount, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"version", nullptr, Version, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"groupCreate", nullptr, GroupCreate, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"groupUpload", nullptr, GroupUpload, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"groupGenerate", nullptr, GroupGenerate, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"groupSetUniforms", nullptr, GroupSetUniforms, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"groupSetModeUniforms", nullptr, GroupSetModeUniforms, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"groupStep", nullptr, GroupStep, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"groupSynchronize", nullptr, GroupSynchronize, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"groupDownloadParticles", nullptr, GroupDownloadParticles, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"groupDownloadField", nullptr, GroupDownloadField, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"groupInfo", nullptr, GroupInfo, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"groupSlabGridI32", nullptr, GroupSlabGridI32, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"rcclSelftest", nullptr, RcclSelftest, nullptr, nullptr, nullptr, napi_default, nullptr},
    };
    napi_define_properties(env, exports, sizeof props / sizeof props[0], props);
    return exports;
}

}  // namespace

NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
