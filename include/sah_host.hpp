// sah_host.hpp — C++17 host façade above the C ABI (sah_hip.h).
//
// Mirrors the reference's pass / render-graph surface for the hot path so that host code reads like
// RenderCore/render/scene_renderer.cpp:365-449:
//     RenderGraph graph{backend};
//     gi->post_render(graph, view, scene, gbuffer, noise);                 // LPV: propagate
//     lighting.render(graph, view, gbuffer, lit_scene, ao, gi, ...);       // -> sah_lighting
//     bloomer.fill_bloom_tex(graph, antialiased);                          // -> sah_bloom
//     ui.render(graph, view, bloomer.get_bloom_tex());                     // -> sah_tonemap
//     graph.finish();
// Names, argument meaning and error behaviour follow the reference (file:line cited per class); resources are linear
// device allocations instead of VkImages.  Header-only; link libsah_hip.so and the HIP runtime.
#pragma once

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <functional>
#include <memory>
#include <optional>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "sah_hip.h"

namespace sah {

// ---- resources (RenderCore/render/backend/handles.hpp:3-5, resource_allocator.hpp) ---------------------------------
struct Texture {
    std::string name;
    sah_volume desc{};      // depth == 1 for plain 2D textures
    sah_mipchain mips{};    // num_mips > 0: a mipped 2D texture (the bloom chain, bloomer.cpp:268-285); desc describes mip 0
    sah_plane plane() const { return sah_plane{desc.ptr, desc.width, desc.height, desc.row_pitch_bytes, desc.format}; }
};
using TextureHandle = Texture*;
// A buffer is host-visible uniform data here (the view and sun UBOs): the C ABI takes those blocks as host pointers
struct Buffer {
    std::string name;
    const void* data = nullptr;
    size_t size = 0;
};
using BufferHandle = const Buffer*;

// RenderCore/render/backend/texture_usage_token.hpp:9-19, buffer_usage_token.hpp: what a pass declares it touches.  The HIP path is
// ordered by its one stream, so the tokens are recorded (tests read them back) but no barrier comes out of them.
struct TextureUsageToken {
    TextureHandle texture = nullptr;
    uint64_t stage = 0, access = 0;
    uint32_t layout = 0;
};
struct BufferUsageToken {
    BufferHandle buffer = nullptr;
    uint64_t stage = 0, access = 0;
};
using TextureUsageList = std::vector<TextureUsageToken>;
using BufferUsageList = std::vector<BufferUsageToken>;

inline uint32_t format_bytes(uint32_t f) {
    switch (f) {
        case SAH_FORMAT_R8_UNORM: return 1;
        case SAH_FORMAT_R16_SFLOAT: case SAH_FORMAT_D16_UNORM: return 2;
        case SAH_FORMAT_R16G16B16A16_SFLOAT: return 8;
        default: return 4;
    }
}

// Creation failures throw std::runtime_error, as ResourceAllocator::create_texture does (resource_allocator.cpp:110-112).
class ResourceAllocator {
public:
    ~ResourceAllocator() {
        for (auto& t : textures) (void)hipFree(t->desc.ptr);
    }
    TextureHandle create_texture(const std::string& name, uint32_t format, uint32_t w, uint32_t h, uint32_t layers = 1) {
        auto t = std::make_unique<Texture>();
        t->name = name;
        const uint32_t pitch = ((w * format_bytes(format) + 255u) / 256u) * 256u;  // 256-byte aligned rows
        t->desc = sah_volume{nullptr, w, h, layers, pitch, pitch * h, format};
        if (hipMalloc(&t->desc.ptr, (size_t)pitch * h * layers) != hipSuccess) throw std::runtime_error("Could not create texture " + name);
        (void)hipMemset(t->desc.ptr, 0, (size_t)pitch * h * layers);
        textures.push_back(std::move(t));
        return textures.back().get();
    }
    TextureHandle create_volume_texture(const std::string& name, uint32_t format, uint32_t w, uint32_t h, uint32_t d) {
        return create_texture(name, format, w, h, d);
    }
    // tightly packed host rows -> device texture
    void upload(TextureHandle t, const void* src, uint32_t src_row_bytes) {
        const size_t rows = (size_t)t->desc.height * t->desc.depth;
        if (hipMemcpy2D(t->desc.ptr, t->desc.row_pitch_bytes, src, src_row_bytes, src_row_bytes, rows, hipMemcpyHostToDevice) != hipSuccess)
            throw std::runtime_error("upload failed: " + t->name);
    }
    void download(TextureHandle t, void* dst, uint32_t dst_row_bytes) {
        const size_t rows = (size_t)t->desc.height * t->desc.depth;
        if (hipMemcpy2D(dst, dst_row_bytes, t->desc.ptr, t->desc.row_pitch_bytes, dst_row_bytes, rows, hipMemcpyDeviceToHost) != hipSuccess)
            throw std::runtime_error("download failed: " + t->name);
    }

private:
    std::vector<std::unique_ptr<Texture>> textures;
};

// RenderCore/render/gbuffer.hpp:5-11
struct GBuffer {
    TextureHandle color = nullptr, normals = nullptr, data = nullptr, emission = nullptr, depth = nullptr;
};

// RenderCore/render/noise_texture.hpp:11-22 (the spatio-temporal blue noise layers; the PNGs are not in the reference tree)
struct NoiseTexture {
    std::vector<TextureHandle> layers;
    uint32_t resolution[2] = {0, 0};
    uint32_t num_layers = 0;
    TextureHandle get_layer(uint32_t index) const { return layers.empty() ? nullptr : layers[index % (uint32_t)layers.size()]; }
};

// ---- backend + immediate render graph (RenderCore/render/backend/render_graph.hpp:24-106, render_pass.hpp:27-130) ----------
class RenderBackend {
public:
    explicit RenderBackend(int device = 0) {
        if (int rc = sah_create(&ctx, device, 0, 1, nullptr); rc != SAH_OK) throw std::runtime_error(std::string("sah_create: ") + sah_status_string(rc));
    }
    ~RenderBackend() { sah_destroy(ctx); }
    RenderBackend(const RenderBackend&) = delete;
    ResourceAllocator& get_global_allocator() { return allocator; }
    sah_ctx* get_context() const { return ctx; }

private:
    sah_ctx* ctx = nullptr;
    ResourceAllocator allocator;
};

// RenderCore/render/backend/texture_descriptor_pool.hpp:15-33 — the bindless texture set the material shaders index
// (`textures[material.base_color_texture_index]`, gltf_basic_pbr.slang:80,177-226).  A slot is a mipped R8G8B8A8 image and the sampler
// it is bound with; the pool keeps the table the rasteriser reads (sah_scene_geometry::textures) in device memory.  The reference passes a
// VkImage with its mip chain and a VkSampler object; here: the levels as textures of the allocator (mip 0 first) and the VkSamplerCreateInfo
// fields the sampling rules of sah_hip.h use.
class TextureDescriptorPool {
public:
    explicit TextureDescriptorPool(RenderBackend& backend_in) : backend(backend_in) {}
    ~TextureDescriptorPool() {
        if (device_table) (void)hipFree(device_table);
    }
    TextureDescriptorPool(const TextureDescriptorPool&) = delete;
    uint32_t create_texture_srv(const std::vector<TextureHandle>& levels, const sah_sampler& sampler) {
        if (levels.empty() || levels.size() > SAH_MAX_TEXTURE_MIPS) throw std::runtime_error("create_texture_srv: 1.." + std::to_string(SAH_MAX_TEXTURE_MIPS) + " levels");
        sah_texture t{};
        for (size_t i = 0; i < levels.size(); i++) t.mips[i] = levels[i]->plane();
        t.num_mips = (uint32_t)levels.size();
        t.sampler = sampler;
        uint32_t handle;
        if (!available_handles.empty()) {
            handle = available_handles.back();
            available_handles.pop_back();
            table[handle] = t;
        } else {
            handle = (uint32_t)table.size();
            table.push_back(t);
        }
        dirty = true;
        return handle;
    }
    void free_descriptor(uint32_t handle) { available_handles.push_back(handle); }
    // "Commits pending descriptor writes.  Should be called at start of frame": uploads the table if it changed
    void commit_descriptors() {
        if (!dirty) return;
        if (device_table) (void)hipFree(device_table);
        device_table = nullptr;
        if (!table.empty()) {
            if (hipMalloc((void**)&device_table, table.size() * sizeof(sah_texture)) != hipSuccess ||
                hipMemcpy(device_table, table.data(), table.size() * sizeof(sah_texture), hipMemcpyHostToDevice) != hipSuccess)
                throw std::runtime_error("texture descriptor table upload failed");
        }
        dirty = false;
    }
    const sah_texture* get_descriptor_set() const { return device_table; }
    uint32_t size() const { return (uint32_t)table.size(); }

private:
    RenderBackend& backend;
    std::vector<sah_texture> table;
    std::vector<uint32_t> available_handles;
    sah_texture* device_table = nullptr;
    bool dirty = false;
};

// RenderCore/render/backend/command_buffer.hpp:48-275, reduced to what the hot path records.  In the reference the sub-passes of
// the "Lighting" render pass are draws that blend into lit_scene one after the other; here they are fused into ONE sah_lighting
// call, so each of them records its part of the sah_lighting_desc into the command buffer (what it would have bound) and the
// phase submits the descriptor once at the end (LightingPhase::render).
class CommandBuffer {
public:
    explicit CommandBuffer(sah_ctx* ctx_in) : ctx(ctx_in) {}
    sah_ctx* get_context() const { return ctx; }
    void begin_label(const std::string&) {}
    void end_label() {}
    // status of the sah_* calls made while recording: a failing call is logged and the frame goes on, which is how the reference
    // treats pipeline failures (pipeline_cache.cpp:166-170)
    void check(int rc, const std::string& what = "") {
        if (rc != SAH_OK) errors.push_back((what.empty() ? current_pass : what) + ": " + sah_status_string(rc) + " (" + sah_last_error(ctx) + ")");
    }
    std::vector<std::string> errors;
    std::string current_pass;

    // the Lighting pass being assembled
    struct LightingRecording {
        bool open = false;
        sah_gbuffer gbuffer{};
        sah_plane lit{}, ao{}, shadow_mask{};
        sah_volume shadowmap{};
        sah_gi gi{};
        sah_sky_luts sky{};
        const sah_view_data* view = nullptr;
        const sah_sun_light_constants* sun = nullptr;
        bool has_ao = false, has_mask = false, has_shadowmap = false, has_gi = false, has_sky = false;
        uint32_t flags = SAH_LIGHTING_DEFAULT_FLAGS;
    } lighting;
    void submit_lighting() {
        if (!lighting.open) return;
        sah_lighting_desc d{};
        d.gbuffer = &lighting.gbuffer;
        d.lit = &lighting.lit;
        d.view = lighting.view;
        d.sun = lighting.sun;
        if (lighting.has_ao) d.ao = &lighting.ao;
        if (lighting.has_mask) d.shadow_mask = &lighting.shadow_mask;
        if (lighting.has_shadowmap) d.shadowmap = &lighting.shadowmap;
        if (lighting.has_gi) d.gi = &lighting.gi;
        if (lighting.has_sky) d.sky = &lighting.sky;
        d.flags = lighting.flags;
        check(sah_lighting(ctx, &d), "Lighting");
        lighting = LightingRecording{};
    }

private:
    sah_ctx* ctx;
};

// render_pass.hpp:27-43
struct ComputePass {
    std::string name;
    TextureUsageList textures;
    BufferUsageList buffers;
    std::function<void(CommandBuffer&)> execute;
};
// A "compute shader" of this backend: what PipelineCache::create_pipeline(path) returns in the reference (pipeline_cache.hpp:23) is a
// handle to a SPIR-V pipeline; here it is a handle to a launcher over the C ABI (push constants, workgroup counts)
struct ComputePipeline {
    std::string name;
    std::function<int(sah_ctx*, const void* push_constants, const uint32_t num_workgroups[3])> launch;
};
using ComputePipelineHandle = const ComputePipeline*;
// descriptor_set_builder.hpp:36-51: what a bound set contributes to a pass is the usage of its resources (the bindings themselves are
// arguments of the C ABI call the pipeline's launcher makes)
struct DescriptorSet {
    TextureUsageList textures;
    BufferUsageList buffers;
    void get_resource_usage_information(TextureUsageList& texture_usages, BufferUsageList& buffer_usages) const {
        texture_usages.insert(texture_usages.end(), textures.begin(), textures.end());
        buffer_usages.insert(buffer_usages.end(), buffers.begin(), buffers.end());
    }
};
// render_pass.hpp:48-79
template <typename PushConstantsType = uint32_t> struct ComputeDispatch {
    std::string name;
    std::vector<DescriptorSet> descriptor_sets;
    BufferUsageList buffers;
    PushConstantsType push_constants{};
    uint32_t num_workgroups[3] = {1, 1, 1};
    ComputePipelineHandle compute_shader = nullptr;
};
// render_pass.hpp:86-116: the workgroup counts come from a buffer (three uint32: VkDispatchIndirectCommand) — host-visible here, like
// every Buffer of this backend, and read when the pass executes
template <typename PushConstantsType = uint32_t> struct IndirectComputeDispatch {
    std::string name;
    std::vector<DescriptorSet> descriptor_sets;
    BufferUsageList buffers;
    PushConstantsType push_constants{};
    BufferHandle dispatch = nullptr;
    ComputePipelineHandle compute_shader = nullptr;
};
// render_pass.hpp:124-130
struct BufferCopyPass {
    std::string name;
    BufferHandle dst = nullptr, src = nullptr;
};
// render_pass.hpp (RenderingAttachmentInfo / DynamicRenderingPass): load_op 1 = VK_ATTACHMENT_LOAD_OP_CLEAR
struct RenderingAttachmentInfo {
    TextureHandle image = nullptr;
    uint32_t load_op = 0;
};
struct DynamicRenderingPass {
    std::string name;
    TextureUsageList textures;
    BufferUsageList buffers;
    std::vector<RenderingAttachmentInfo> color_attachments;
    std::function<void(CommandBuffer&)> execute;
};
struct TransitionPass {
    TextureUsageList textures;
    BufferUsageList buffers;
};
struct ImageCopyPass {
    std::string name;
    TextureHandle dst = nullptr, src = nullptr;
};
// A pass written as "status = one sah_* call": the form most passes of this file take
template <class F> ComputePass hip_pass(std::string name, F fn) {
    ComputePass p;
    p.name = std::move(name);
    p.execute = [fn](CommandBuffer& commands) { commands.check(fn(commands.get_context())); };
    return p;
}

// Passes execute immediately, in the order they are added (render_graph.cpp:85-111,113-222): the graph owns one command buffer and
// hands it to every pass.
class RenderGraph {
public:
    explicit RenderGraph(RenderBackend& backend_in) : backend(backend_in), cmds(backend_in.get_context()) {}
    void add_transition_pass(const TransitionPass& pass) { record_usages(pass.textures, pass.buffers); }  // one stream: ordering is implicit
    void add_copy_pass(const ImageCopyPass& pass) {  // mip 0 of one image to mip 0 of the other (render_graph.hpp:41-44)
        cmds.current_pass = pass.name;
        const sah_volume &d = pass.dst->desc, &s = pass.src->desc;
        if (d.width != s.width || d.height != s.height || d.format != s.format) {
            cmds.errors.push_back(pass.name + ": image copy needs matching extents and formats");
            return;
        }
        // the context's stream is not visible through the C ABI: synchronise, then copy on the null stream
        cmds.check(sah_sync(backend.get_context()));
        if (hipMemcpy2D(d.ptr, d.row_pitch_bytes, s.ptr, s.row_pitch_bytes, (size_t)s.width * format_bytes(s.format), (size_t)s.height * s.depth,
                        hipMemcpyDeviceToDevice) != hipSuccess)
            cmds.errors.push_back(pass.name + ": hipMemcpy2D failed");
    }
    void add_copy_pass(const BufferCopyPass& pass) {  // render_graph.hpp:36-39; buffers are host-visible blocks here
        cmds.current_pass = pass.name;
        if (!pass.dst || !pass.src || !pass.dst->data || !pass.src->data || pass.dst->size < pass.src->size) {
            cmds.errors.push_back(pass.name + ": buffer copy needs a destination at least as large as the source");
            return;
        }
        cmds.check(sah_sync(backend.get_context()));  // work enqueued earlier may still read the destination
        std::memcpy(const_cast<void*>(pass.dst->data), pass.src->data, pass.src->size);
        num_passes++;
    }
    void add_pass(ComputePass pass) {
        cmds.current_pass = pass.name;
        record_usages(pass.textures, pass.buffers);
        if (pass.execute) pass.execute(cmds);
        num_passes++;
    }
    template <typename PushConstantsType = uint32_t> void add_compute_dispatch(const ComputeDispatch<PushConstantsType>& dispatch_info) {
        cmds.current_pass = dispatch_info.name;
        record_sets(dispatch_info.descriptor_sets);
        record_usages({}, dispatch_info.buffers);
        if (!dispatch_info.compute_shader) {
            cmds.errors.push_back(dispatch_info.name + ": null compute shader");  // the reference logs a null pipeline and goes on
            return;
        }
        cmds.check(dispatch_info.compute_shader->launch(backend.get_context(), &dispatch_info.push_constants, dispatch_info.num_workgroups));
        num_passes++;
    }
    template <typename PushConstantsType = uint32_t> void add_compute_dispatch(const IndirectComputeDispatch<PushConstantsType>& dispatch_info) {
        cmds.current_pass = dispatch_info.name;
        record_sets(dispatch_info.descriptor_sets);
        record_usages({}, dispatch_info.buffers);
        if (!dispatch_info.compute_shader || !dispatch_info.dispatch || !dispatch_info.dispatch->data || dispatch_info.dispatch->size < 3 * sizeof(uint32_t)) {
            cmds.errors.push_back(dispatch_info.name + ": indirect dispatch needs a compute shader and a buffer of three workgroup counts");
            return;
        }
        uint32_t num_workgroups[3];
        std::memcpy(num_workgroups, dispatch_info.dispatch->data, sizeof(num_workgroups));
        cmds.check(dispatch_info.compute_shader->launch(backend.get_context(), &dispatch_info.push_constants, num_workgroups));
        num_passes++;
    }
    void add_render_pass(DynamicRenderingPass pass) {
        cmds.current_pass = pass.name;
        record_usages(pass.textures, pass.buffers);
        if (pass.execute) pass.execute(cmds);
        num_passes++;
    }
    void begin_label(const std::string&) {}
    void end_label() {}
    void finish() {
        if (int rc = sah_sync(backend.get_context()); rc != SAH_OK) cmds.errors.push_back(std::string("finish: ") + sah_status_string(rc));
    }
    const std::vector<std::string>& get_errors() const { return cmds.errors; }
    const TextureUsageList& get_texture_usages() const { return texture_usages; }
    uint32_t get_num_passes() const { return num_passes; }

private:
    void record_sets(const std::vector<DescriptorSet>& sets) {
        for (const auto& set : sets) set.get_resource_usage_information(texture_usages, buffer_usages);
    }
    void record_usages(const TextureUsageList& textures, const BufferUsageList& buffers) {
        texture_usages.insert(texture_usages.end(), textures.begin(), textures.end());
        buffer_usages.insert(buffer_usages.end(), buffers.begin(), buffers.end());
    }
    RenderBackend& backend;
    CommandBuffer cmds;
    TextureUsageList texture_usages;
    BufferUsageList buffer_usages;
    uint32_t num_passes = 0;
};

// ---- small column-major matrix helpers (glm stand-ins; inputs to the path, SURVEY §8 "types") -------------------------
using Mat4 = std::array<float, 16>;  // m[col*4 + row]
using Vec3 = std::array<float, 3>;

inline Mat4 mat_identity() { return Mat4{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}; }
inline Mat4 mat_mul(const Mat4& a, const Mat4& b) {
    Mat4 r{};
    for (int j = 0; j < 4; j++)
        for (int i = 0; i < 4; i++) {
            float s = 0.f;
            for (int k = 0; k < 4; k++) s += a[k * 4 + i] * b[j * 4 + k];
            r[j * 4 + i] = s;
        }
    return r;
}
inline Mat4 mat_inverse(const Mat4& m) {  // Gauss-Jordan in double, rounded to fp32
    double a[4][8];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            a[i][j] = m[j * 4 + i];
            a[i][4 + j] = i == j ? 1.0 : 0.0;
        }
    for (int c = 0; c < 4; c++) {
        int piv = c;
        for (int r = c + 1; r < 4; r++)
            if (std::fabs(a[r][c]) > std::fabs(a[piv][c])) piv = r;
        for (int j = 0; j < 8; j++) std::swap(a[c][j], a[piv][j]);
        const double d = a[c][c];
        for (int j = 0; j < 8; j++) a[c][j] /= d;
        for (int r = 0; r < 4; r++)
            if (r != c) {
                const double f = a[r][c];
                for (int j = 0; j < 8; j++) a[r][j] -= f * a[c][j];
            }
    }
    Mat4 r{};
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) r[j * 4 + i] = (float)a[i][4 + j];
    return r;
}
inline Vec3 v_sub(Vec3 a, Vec3 b) { return {a[0] - b[0], a[1] - b[1], a[2] - b[2]}; }
inline Vec3 v_cross(Vec3 a, Vec3 b) { return {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]}; }
inline float v_dot(Vec3 a, Vec3 b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline Vec3 v_normalize(Vec3 a) {
    const float inv = 1.0f / std::sqrt(v_dot(a, a));
    return {a[0] * inv, a[1] * inv, a[2] * inv};
}
inline Mat4 ortho(float left, float right, float bottom, float top, float z_near, float z_far) {  // glm::ortho, RH, depth 0..1
    Mat4 m{};
    m[0] = 2.f / (right - left); m[5] = 2.f / (top - bottom); m[10] = -1.f / (z_far - z_near);
    m[12] = -(right + left) / (right - left); m[13] = -(top + bottom) / (top - bottom); m[14] = -z_near / (z_far - z_near); m[15] = 1.f;
    return m;
}
inline Mat4 perspective_fov(float fov, float width, float height, float z_near, float z_far) {  // glm::perspectiveFov, RH, depth 0..1
    const float h = std::cos(0.5f * fov) / std::sin(0.5f * fov), w = h * height / width;
    Mat4 m{};
    m[0] = w; m[5] = h; m[10] = z_far / (z_near - z_far); m[11] = -1.f; m[14] = -(z_far * z_near) / (z_far - z_near);
    return m;
}
inline Mat4 look_at(Vec3 eye, Vec3 center, Vec3 up) {  // glm::lookAt, right-handed
    const Vec3 f = v_normalize(v_sub(center, eye)), s = v_normalize(v_cross(f, up)), u = v_cross(s, f);
    Mat4 m = mat_identity();
    m[0] = s[0]; m[4] = s[1]; m[8] = s[2];
    m[1] = u[0]; m[5] = u[1]; m[9] = u[2];
    m[2] = -f[0]; m[6] = -f[1]; m[10] = -f[2];
    m[12] = -v_dot(s, eye); m[13] = -v_dot(u, eye); m[14] = v_dot(f, eye);
    return m;
}

// ---- SceneView (RenderCore/render/scene_view.cpp:13-27,138-187) --------------------------------------------------------
class SceneView {
public:
    SceneView() = default;
    SceneView(const SceneView&) = delete;  // `buffer` points at this object's gpu_data
    SceneView& operator=(const SceneView&) = delete;
    void set_render_resolution(uint32_t w, uint32_t h) { gpu_data.render_resolution[0] = (float)w; gpu_data.render_resolution[1] = (float)h; }
    void set_position(Vec3 p) { position = p; }
    void rotate(float delta_pitch, float delta_yaw) { pitch += delta_pitch; yaw += delta_yaw; }
    void set_perspective_projection(float fov_in, float aspect_in, float near_in) { fov = fov_in; aspect = aspect_in; near_value = near_in; }
    Vec3 get_position() const { return position; }
    Vec3 get_forward() const { return forward; }
    float get_fov() const { return fov; }  // degrees (scene_view.cpp:95-97)
    float get_aspect_ratio() const { return aspect; }
    float get_near() const { return near_value; }
    const sah_view_data& get_gpu_data() const { return gpu_data; }
    BufferHandle get_buffer() const { return &buffer; }  // scene_view.hpp: the ViewDataGPU uniform buffer
    void update_transforms() {
        forward = {std::cos(pitch) * std::sin(yaw), std::sin(pitch), std::cos(pitch) * std::cos(yaw)};
        const float half_pi = 3.14159265358979f / 2.0f;
        const Vec3 right = {std::sin(yaw - half_pi), 0.f, std::cos(yaw - half_pi)};
        const Vec3 up = v_cross(right, forward);
        const Mat4 view = look_at(position, {position[0] + forward[0], position[1] + forward[1], position[2] + forward[2]}, up);
        std::memcpy(gpu_data.last_frame_view, gpu_data.view, 64);
        std::memcpy(gpu_data.view, view.data(), 64);
        std::memcpy(gpu_data.inverse_view, mat_inverse(view).data(), 64);
        // inf_depth_reverse_z_perspective, scene_view.cpp:13-27
        const float t = 1.0f / std::tan(fov * 3.14159265358979f / 180.0f * 0.5f);
        Mat4 proj{};
        proj[0] = t / aspect; proj[5] = t; proj[11] = -1.0f; proj[14] = near_value;
        std::memcpy(gpu_data.last_frame_projection, gpu_data.projection, 64);
        std::memcpy(gpu_data.projection, proj.data(), 64);
        std::memcpy(gpu_data.inverse_projection, mat_inverse(proj).data(), 64);
        gpu_data.z_near = near_value;
    }

private:
    float fov = 75.f, aspect = 16.f / 9.f, near_value = 0.05f, pitch = 0.f, yaw = 0.f;
    Vec3 position{}, forward{};
    sah_view_data gpu_data{};
    Buffer buffer{"view data", &gpu_data, sizeof(sah_view_data)};
};

// ---- DirectionalLight (RenderCore/render/directional_light.cpp:232-260, render_scene.cpp:25-27) ------------------------
enum class SunShadowMode : uint32_t { Off = 0, CascadedShadowMaps = 1, RayTracing = 2 };
class DirectionalLight {
public:
    DirectionalLight() {
        set_direction({0.1f, -1.f, -1.f});
        constants.color[0] = constants.color[1] = constants.color[2] = 80000.f;
        constants.shadow_mode = (uint32_t)SunShadowMode::RayTracing;  // r.Shadow.SunShadowMode default
        constants.num_shadow_samples = 8.f;
    }
    void set_direction(Vec3 d) {
        const Vec3 n = v_normalize(d);
        for (int i = 0; i < 3; i++) constants.direction_and_tan_size[i] = n[i];
        constants.direction_and_tan_size[3] = std::tan(0.545f * 3.14159265358979f / 180.0f);
    }
    // directional_light.cpp:84-230: practical split scheme (lambda blend of logarithmic and uniform splits), a bounding sphere per
    // slice of the camera frustum, radius doubled and snapped to 1/16, an orthographic light frustum looking at the sphere centre.
    // glm::perspectiveFov receives get_fov() — degrees — where it expects radians, as in the reference (:164-170).
    void update_shadow_cascades(const SceneView& view, uint32_t num_cascades = 4, float max_shadow_distance = 128.f, float cascade_split_lambda = 0.95f,
                                uint32_t csm_resolution = 4096) {
        if (get_shadow_mode() != SunShadowMode::CascadedShadowMaps) return;
        const float z_near = view.get_near(), clip_range = z_near + max_shadow_distance, ratio = clip_range / z_near;
        float splits[4] = {};
        for (uint32_t i = 0; i < num_cascades && i < 4; i++) {
            const float p = (float)((int32_t)i + 1) / (float)num_cascades;
            const float log = z_near * std::pow(ratio, p), uniform = z_near + max_shadow_distance * p;
            splits[i] = (cascade_split_lambda * (log - uniform) + uniform - z_near) / clip_range;
        }
        const Vec3 light_dir = v_normalize({constants.direction_and_tan_size[0], constants.direction_and_tan_size[1], constants.direction_and_tan_size[2]});
        float last_split = z_near;
        for (uint32_t i = 0; i < num_cascades && i < 4; i++) {
            const Mat4 projection = perspective_fov(view.get_fov(), view.get_aspect_ratio(), 1.f, last_split * max_shadow_distance, splits[i] * max_shadow_distance);
            Mat4 view_matrix;
            std::memcpy(view_matrix.data(), view.get_gpu_data().view, 64);
            const Mat4 inverse_camera = mat_inverse(mat_mul(projection, view_matrix));
            Vec3 corners[8];
            Vec3 center{};
            for (int c = 0; c < 8; c++) {
                const float ndc[4] = {(c & 1) ^ ((c >> 1) & 1) ? 1.f : -1.f, (c & 2) ? -1.f : 1.f, (c & 4) ? 1.f : -1.f, 1.f};
                float w[4] = {};
                for (int r = 0; r < 4; r++)
                    for (int k = 0; k < 4; k++) w[r] += inverse_camera[k * 4 + r] * ndc[k];
                corners[c] = {w[0] / w[3], w[1] / w[3], w[2] / w[3]};
                for (int k = 0; k < 3; k++) center[k] += corners[c][k];
            }
            for (int k = 0; k < 3; k++) center[k] /= 8.f;
            float radius = 0.f;
            for (const Vec3& c : corners) radius = std::max(radius, std::sqrt(v_dot(v_sub(c, center), v_sub(c, center))));
            radius = std::ceil(radius * 2.f * 16.f) / 16.f;
            const Mat4 light_view = look_at({center[0] - light_dir[0] * radius, center[1] - light_dir[1] * radius, center[2] - light_dir[2] * radius}, center, {0.f, 1.f, 0.f});
            const Mat4 m = mat_mul(ortho(-radius, radius, -radius, radius, 0.f, radius + radius), light_view);
            for (int k = 0; k < 4; k++) constants.data[i][k] = 0.f;
            constants.data[i][0] = splits[i] * clip_range * -1.f;
            std::memcpy(constants.cascade_matrices[i], m.data(), 64);
            std::memcpy(constants.cascade_inverse_matrices[i], mat_inverse(m).data(), 64);
            last_split = splits[i];
        }
        constants.csm_resolution[0] = constants.csm_resolution[1] = csm_resolution;
    }
    // directional_light.cpp:286-327: the "Sun shadow" pass — every primitive into every cascade layer, depth only
    void render_shadows(RenderGraph& graph, const sah_scene_geometry& geometry, uint32_t num_cascades = 4) const {
        if (get_shadow_mode() != SunShadowMode::CascadedShadowMaps || !shadowmap_handle) return;
        graph.add_pass(hip_pass("Sun shadow", [this, &geometry, num_cascades](sah_ctx* ctx) {
                            return sah_shadow_render(ctx, &geometry, &constants, num_cascades, &shadowmap_handle->desc, nullptr);
                        }));
    }
    // directional_light.cpp:329-370: the CSM-mode sun draw inside the Lighting pass — here: what it binds goes into the recording
    void render(CommandBuffer& commands, const SceneView& view) const {
        (void)view;
        commands.lighting.sun = &constants;
        if (shadowmap_handle) {
            commands.lighting.shadowmap = shadowmap_handle->desc;
            commands.lighting.has_shadowmap = true;
        }
    }
    // directional_light.cpp:372-422: the RT-mode sun dispatch after the Lighting pass.  The shadow rays of the raygen shader
    // (directional_light.rt.slang:91-125) are traced here into `shadow_mask` (sah_sun_shadow_mask, against the structure
    // RaytracingScene::finalize built) whenever a noise layer is given; the shading half of that shader joins the fused Lighting call.
    // Without a noise layer `shadow_mask` is taken as it is (a mask made elsewhere).
    void raytrace(RenderGraph& graph, const SceneView& view, const GBuffer& gbuffer, const struct RenderScene& scene, TextureHandle lit_scene,
                  const NoiseTexture& noise) {
        (void)scene; (void)lit_scene;
        if (TextureHandle layer = noise.get_layer(frame_index); layer && shadow_mask && gbuffer.depth && gbuffer.normals) {
            graph.add_pass(hip_pass("ray_traced_sun: shadow rays", [this, &view, &gbuffer, layer](sah_ctx* ctx) {
                                const sah_plane d = gbuffer.depth->plane(), n = gbuffer.normals->plane(), z = layer->plane(), m = shadow_mask->plane();
                                return sah_sun_shadow_mask(ctx, &view.get_gpu_data(), &constants, &d, &n, &z, &m);
                            }));
        }
        frame_index++;  // :392
        ComputePass pass;
        pass.name = "Raytraced sun (recorded into the Lighting call)";
        pass.execute = [this](CommandBuffer& commands) {
            commands.lighting.sun = &constants;
            if (shadow_mask) {
                commands.lighting.shadow_mask = shadow_mask->plane();
                commands.lighting.has_mask = true;
            }
        };
        graph.add_pass(std::move(pass));
    }
    BufferHandle get_constant_buffer() const { return &buffer; }
    void set_shadow_mode(SunShadowMode m) { constants.shadow_mode = (uint32_t)m; }
    SunShadowMode get_shadow_mode() const { return (SunShadowMode)constants.shadow_mode; }
    sah_sun_light_constants& get_constants() { return constants; }
    const sah_sun_light_constants& get_constants() const { return constants; }
    TextureHandle shadowmap_handle = nullptr;  // D16 array, CSM mode
    TextureHandle shadow_mask = nullptr;       // RT mode: visibility fraction, shadow / num_shadow_samples (R32_SFLOAT)
    uint32_t frame_index = 0;                  // selects the noise layer (directional_light.hpp)

    DirectionalLight(const DirectionalLight&) = delete;  // `buffer` points at this object's constants
    DirectionalLight& operator=(const DirectionalLight&) = delete;

private:
    sah_sun_light_constants constants{};
    Buffer buffer{"sun constants", &constants, sizeof(sah_sun_light_constants)};
};

// RenderCore/render/procedural_sky.{hpp,cpp}: LUT allocation :31-60, update_sky_luts :75-149; the sky fill reads two of the LUTs
struct ProceduralSky {
    TextureHandle transmittance_lut = nullptr, multiscattering_lut = nullptr, sky_view_lut = nullptr;
    void create_luts(ResourceAllocator& alloc) {
        transmittance_lut = alloc.create_texture("Transmittance LUT", SAH_FORMAT_R16G16B16A16_SFLOAT, 256, 64);
        multiscattering_lut = alloc.create_texture("Multiscattering LUT", SAH_FORMAT_R16G16B16A16_SFLOAT, 32, 32);
        sky_view_lut = alloc.create_texture("Sky view LUT", SAH_FORMAT_R16G16B16A16_SFLOAT, 200, 200);
    }
    // procedural_sky.cpp:151-172: the sky fill inside the Lighting pass (depth == 0 pixels)
    void render_sky(CommandBuffer& commands, BufferHandle view_buffer, BufferHandle sun_buffer, TextureHandle depth_buffer) const {
        (void)view_buffer; (void)depth_buffer;
        if (!sky_view_lut || !transmittance_lut) return;
        commands.lighting.sky = sah_sky_luts{transmittance_lut->plane(), sky_view_lut->plane()};
        commands.lighting.has_sky = true;
        if (sun_buffer && sun_buffer->data && !commands.lighting.sun) commands.lighting.sun = static_cast<const sah_sun_light_constants*>(sun_buffer->data);
    }
    void update_sky_luts(RenderGraph& graph, const Vec3& light_vector) const {
        graph.add_pass(hip_pass("Update sky LUTs", [this, light_vector](sah_ctx* ctx) {
                            const sah_plane t = transmittance_lut->plane(), m = multiscattering_lut->plane(), s = sky_view_lut->plane();
                            return sah_sky_update_luts(ctx, &t, &m, &s, light_vector.data());
                        }));
    }
};

// RenderCore/render/raytracing_scene.{hpp,cpp}: the TLAS over one instance per primitive.  add_primitive marks the scene dirty
// (:15-43: the instance's transform, mask and opacity flags are already in sah_primitive), finalize() commits the pending build
// (:45-170) — here one sah_rt_build over the scene's geometry pool.
class RaytracingScene {
public:
    explicit RaytracingScene(struct RenderScene& scene_in) : scene(scene_in) {}
    void add_primitive(uint32_t /*primitive index in scene.geometry.primitives*/) { is_dirty = true; }
    void finalize(RenderGraph& graph);
    bool is_built() const { return built; }

private:
    struct RenderScene& scene;
    bool is_dirty = false, built = false;
};

struct RenderScene {  // the slice of RenderCore/render/render_scene.hpp the hot path touches
    DirectionalLight sun;
    ProceduralSky sky;
    sah_scene_geometry geometry{};  // device-side mesh pool + primitive buffer (render_scene.hpp: get_meshes(), get_primitive_buffer())
    RaytracingScene raytracing_scene{*this};
    DirectionalLight& get_sun_light() { return sun; }
    ProceduralSky& get_sky() { return sky; }
    RaytracingScene& get_raytracing_scene() { return raytracing_scene; }
    const RaytracingScene& get_raytracing_scene() const { return raytracing_scene; }
};

inline void RaytracingScene::finalize(RenderGraph& graph) {
    if (!is_dirty) return;  // commit_tlas_builds: nothing to do
    graph.add_pass(hip_pass("Build TLAS", [this](sah_ctx* ctx) { return sah_rt_build(ctx, &scene.geometry, nullptr); }));
    is_dirty = false;
    built = true;
}

// ---- GI plugin seam (RenderCore/render/gi/global_illuminator.hpp:18-45): the five methods, with the reference's signatures --------
class IGlobalIlluminator {
public:
    virtual ~IGlobalIlluminator() = default;
    virtual void pre_render(RenderGraph& graph, const SceneView& view, const RenderScene& scene, TextureHandle noise_tex) = 0;
    virtual void post_render(RenderGraph& graph, const SceneView& view, const RenderScene& scene, const GBuffer& gbuffer, TextureHandle noise_tex) = 0;
    virtual void get_lighting_resource_usages(TextureUsageList& textures, BufferUsageList& buffers) const = 0;
    // Runs inside the Lighting pass with the G-buffer already bound (lighting_phase.cpp:109,116).  The overlay of the reference binds
    // its descriptor set and draws a fullscreen triangle; here it records the same bindings into the fused Lighting call.
    virtual void render_to_lit_scene(CommandBuffer& commands, BufferHandle view_buffer, TextureHandle ao_tex, TextureHandle noise_tex) const = 0;
    virtual void draw_debug_overlays(RenderGraph& graph, const SceneView& view, const GBuffer& gbuffer, TextureHandle lit_scene_texture) = 0;
};
constexpr uint64_t kStageFragmentShader = 0x80ull, kAccessShaderRead = 0x20ull;  // VK_PIPELINE_STAGE_2_FRAGMENT_SHADER_BIT, VK_ACCESS_2_SHADER_READ_BIT
constexpr uint32_t kLayoutShaderReadOnly = 5;                                    // VK_IMAGE_LAYOUT_SHADER_READ_ONLY_OPTIMAL

// RenderCore/render/gi/light_propagation_volume.{hpp,cpp}: volumes :321-378, cascades :455-546, clear :839-926,
// propagate :970-1063, overlay :274-306, RSM + VPL extraction + injection :548-760.
class LightPropagationVolume : public IGlobalIlluminator {
public:
    explicit LightPropagationVolume(RenderBackend& backend, uint32_t cascades = 4, uint32_t steps = 32) : num_cascades(cascades), num_steps(steps) {
        auto& alloc = backend.get_global_allocator();
        const char* names[7] = {"LPV Red A", "LPV Green A", "LPV Blue A", "LPV Red B", "LPV Green B", "LPV Blue B", "Geometry Volume"};
        for (int i = 0; i < 7; i++) vol[i] = alloc.create_volume_texture(names[i], SAH_FORMAT_R16G16B16A16_SFLOAT, 32 * cascades, 32, 32);
        // RSM targets and VPL lists, light_propagation_volume.cpp:385-452 (r.GI.LPV.RsmResolution = 128: 4096 lights per cascade)
        rsm_flux = alloc.create_texture("RSM Flux", SAH_FORMAT_R8G8B8A8_SRGB, rsm_resolution, rsm_resolution, cascades);
        rsm_normals = alloc.create_texture("RSM Normals", SAH_FORMAT_R8G8B8A8_UNORM, rsm_resolution, rsm_resolution, cascades);
        rsm_depth = alloc.create_texture("RSM Depth", SAH_FORMAT_D16_UNORM, rsm_resolution, rsm_resolution, cascades);
        const size_t num_vpls = (size_t)rsm_resolution * rsm_resolution / 4;
        if (hipMalloc((void**)&vpl_lists, cascades * num_vpls * sizeof(sah_packed_vpl)) != hipSuccess || hipMalloc((void**)&vpl_counts, cascades * sizeof(uint32_t)) != hipSuccess)
            throw std::runtime_error("LightPropagationVolume: VPL buffers");
    }
    ~LightPropagationVolume() {
        if (vpl_lists) (void)hipFree(vpl_lists);
        if (vpl_counts) (void)hipFree(vpl_counts);
    }
    LightPropagationVolume(const LightPropagationVolume&) = delete;  // owns the VPL buffers
    LightPropagationVolume& operator=(const LightPropagationVolume&) = delete;
    // light_propagation_volume.cpp:548-697: render the RSM of every cascade, extract the VPLs, add them to the A volumes
    void inject_indirect_sun_light(RenderGraph& graph, const RenderScene& scene) {
        graph.add_pass(hip_pass("Render RSM", [this, &scene](sah_ctx* ctx) {
                            const sah_rsm_targets rsm = {rsm_flux->desc, rsm_normals->desc, rsm_depth->desc};
                            return sah_rsm_render(ctx, &scene.geometry, &scene.sun.get_constants(), cascades.data(), num_cascades, &rsm, nullptr);
                        }));
        for (uint32_t c = 0; c < num_cascades; c++)
            graph.add_pass(hip_pass("Extract and inject VPLs", [this, c](sah_ctx* ctx) {
                                const sah_rsm_targets rsm = {rsm_flux->desc, rsm_normals->desc, rsm_depth->desc};
                                const uint32_t num_vpls = rsm_resolution * rsm_resolution / 4;
                                sah_packed_vpl* list = vpl_lists + (size_t)c * num_vpls;
                                if (int rc = sah_lpv_extract_vpls(ctx, &rsm, cascades.data(), c, 0.25f, list, vpl_counts + c); rc != SAH_OK) return rc;
                                const sah_volume a[3] = {vol[0]->desc, vol[1]->desc, vol[2]->desc};
                                return sah_lpv_inject_vpls(ctx, list, vpl_counts + c, num_vpls, cascades.data(), c, num_cascades, a);
                            }));
    }
    void update_cascade_transforms(const SceneView& view, const DirectionalLight& light) {
        const auto& dir = light.get_constants().direction_and_tan_size;
        const Vec3 light_dir = v_normalize({dir[0], dir[1], dir[2]});
        for (uint32_t c = 0; c < num_cascades; c++) {
            const float cell = 0.25f * std::pow(2.f, (float)c), size = 32.f * cell;
            const Vec3 p = view.get_position(), f = view.get_forward();
            Mat4 m = mat_identity();
            for (int i = 0; i < 3; i++) {
                const float off = p[i] + f[i] * (size * (0.5f - 0.1f));
                const float snapped = std::round(off / (cell * 2.f)) * cell * 2.f;
                m[i * 5] = 0.5f * (1.0f / size);                          // bias(0.5) * scale(1/size)
                m[12 + i] = 0.5f * ((1.0f / size) * -snapped) + 0.5f;     // bias * scale * translate(-snapped) + 0.5
            }
            std::memcpy(cascades[c].world_to_cascade, m.data(), 64);
            std::memcpy(cascades[c].cascade_to_world, mat_inverse(m).data(), 64);
            // RSM frustum (:499-531): look at the snapped cascade centre from two cascade sizes up the light direction
            Vec3 centre{};
            for (int i = 0; i < 3; i++) centre[i] = std::round((p[i] + f[i] * (size * (0.5f - 0.1f))) / (cell * 2.f)) * cell * 2.f;
            const float pull = size * 2.f, half = size / 2.f;
            const Mat4 rsm_view = look_at({centre[0] - light_dir[0] * pull, centre[1] - light_dir[1] * pull, centre[2] - light_dir[2] * pull}, centre, {0.f, 1.f, 0.f});
            const Mat4 rsm_vp = mat_mul(ortho(-half, half, -half, half, 0.f, pull * 2.f), rsm_view);
            std::memcpy(cascades[c].rsm_vp, rsm_vp.data(), 64);
            std::memcpy(cascades[c].inverse_rsm_vp, mat_inverse(rsm_vp).data(), 64);
        }
    }
    void pre_render(RenderGraph& graph, const SceneView&, const RenderScene&, TextureHandle) override {
        graph.add_pass(hip_pass("LPV clear", [this](sah_ctx* ctx) {
                            return sah_lpv_clear(ctx, &vol[0]->desc, &vol[1]->desc, &vol[2]->desc, &vol[6]->desc, num_cascades);
                        }));
    }
    void post_render(RenderGraph& graph, const SceneView&, const RenderScene&, const GBuffer&, TextureHandle) override {
        graph.add_pass(hip_pass("LPV Propagation", [this](sah_ctx* ctx) {
                            const sah_volume a[3] = {vol[0]->desc, vol[1]->desc, vol[2]->desc}, b[3] = {vol[3]->desc, vol[4]->desc, vol[5]->desc};
                            return sah_lpv_propagate(ctx, a, b, num_cascades, num_steps);
                        }));
    }
    void get_lighting_resource_usages(TextureUsageList& textures, BufferUsageList&) const override {  // light_propagation_volume.cpp:245-272
        for (int c = 0; c < 3; c++) textures.push_back({vol[c], kStageFragmentShader, kAccessShaderRead, kLayoutShaderReadOnly});
    }
    // to be called by anything that writes the volumes behind the library's back (an upload, another API): sah_gi::lpv_generation
    void mark_volumes_written() { generation = generation == 0xffffffffu ? 1u : generation + 1u; }
    uint32_t generation = 1;
    void render_to_lit_scene(CommandBuffer& commands, BufferHandle view_buffer, TextureHandle ao_tex, TextureHandle) const override {
        (void)view_buffer;  // the view block of the Lighting pass is the one the overlay reads
        sah_gi& gi = commands.lighting.gi;
        gi = sah_gi{};
        gi.kind = SAH_GI_LPV;
        gi.lpv_red = vol[0]->desc; gi.lpv_green = vol[1]->desc; gi.lpv_blue = vol[2]->desc;
        gi.lpv_cascades = cascades.data();
        gi.lpv_num_cascades = num_cascades;
        gi.lpv_exposure = 3.1415927f * 10.f;  // r.GI.LPV.Exposure, pushed as a constant (:298-299)
        gi.lpv_generation = generation;       // the library's own writers (clear / inject / propagate) drop its gather copy themselves
        commands.lighting.has_gi = true;
        if (ao_tex) {  // the LPV overlay is the one consumer of the AO texture (gi/lpv/overlay.frag:153-155)
            commands.lighting.ao = ao_tex->plane();
            commands.lighting.has_ao = true;
        }
    }
    void draw_debug_overlays(RenderGraph&, const SceneView&, const GBuffer&, TextureHandle) override {}  // GV / VPL visualisers: debug views, not on the path
    const sah_lpv_cascade_matrices* get_cascade_matrices() const { return cascades.data(); }
    TextureHandle get_volume(int channel, bool b = false) const { return vol[channel + (b ? 3 : 0)]; }

private:
    uint32_t num_cascades, num_steps;
    uint32_t rsm_resolution = 128;
    TextureHandle vol[7]{};
    TextureHandle rsm_flux = nullptr, rsm_normals = nullptr, rsm_depth = nullptr;
    sah_packed_vpl* vpl_lists = nullptr;
    uint32_t* vpl_counts = nullptr;
    std::array<sah_lpv_cascade_matrices, 4> cascades{};
};

// Probe scheduling of the irradiance cache (CPU side) — RenderCore/render/gi/irradiance_cache.hpp:49-105,267-279 and
// irradiance_cache.cpp:36-60 (grid layout), :351-360 (request_probe_update), :496-583 (find_probes_to_update).  No GPU state: the
// list it builds is what sah_probe_update consumes.  The draw order of the reference is kept — foreach() runs x, then y, then z,
// innermost — and so is its random source: `std::default_random_engine{frame_count}` with `std::uniform_real_distribution<float>`,
// i.e. whatever the C++ standard library of the build defines (minstd_rand0 with libstdc++ and libc++'s minstd_rand give different
// streams, MSVC uses mt19937): built like the reference, it draws like the reference.  Quirk kept: find_probes_to_update requests
// the cascade-local index, without the `y + 8 * cascade` offset that place_probes_from_view adds (:405-406 vs :533).
class ProbeScheduler {
public:
    static constexpr uint32_t cascade_size_xz = 32, cascade_size_y = 8, num_cascades = 4;
    struct Probe {
        bool is_valid = false;
        uint32_t last_update_frame = 0;
    };
    struct Cascade {
        float update_priority = 0.1f;
        std::array<Probe, cascade_size_xz * cascade_size_y * cascade_size_xz> probes{};
        Probe& at(uint32_t x, uint32_t y, uint32_t z) { return probes[x + y * cascade_size_xz + z * cascade_size_xz * cascade_size_y]; }
    };
    using ProbeIndex = std::array<uint32_t, 3>;

    explicit ProbeScheduler(uint32_t probes_per_frame = 1024) : update_budget(probes_per_frame) { probes_to_update.reserve(probes_per_frame); }
    Cascade& cascade(uint32_t c) { return cascades[c]; }
    const std::vector<ProbeIndex>& get_probes_to_update() const { return probes_to_update; }
    void clear_probes_to_update() { probes_to_update.clear(); }  // after dispatch_probe_updates (:722)

    bool request_probe_update(ProbeIndex probe_index) {
        if (probes_to_update.size() == update_budget) return false;
        probes_to_update.push_back(probe_index);
        return true;
    }

    void find_probes_to_update(uint32_t frame_count) {
        if (probes_to_update.size() >= update_budget) return;
        auto rng = std::default_random_engine{frame_count};
        auto distribution = std::uniform_real_distribution<float>{0.f, 1.f};
        float total_weight = 0.f;
        for (const auto& c : cascades) total_weight += c.update_priority;
        // probes that were never traced or were invalidated
        for (auto& c : cascades) {
            const float normalized_priority = c.update_priority / total_weight;
            foreach (c, [&](ProbeIndex index, Probe& probe) {
                if (!probe.is_valid) {
                    const float num = distribution(rng);
                    if (num < normalized_priority) {
                        if (!request_probe_update(index)) return false;
                        probe.is_valid = true;
                        probe.last_update_frame = frame_count;
                    }
                }
                return true;
            });
        }
        if (probes_to_update.size() >= update_budget) return;
        // probes that have not been updated in a while: log(seconds since the update) * priority
        for (auto& c : cascades) {
            const float normalized_priority = c.update_priority / total_weight;
            foreach (c, [&](ProbeIndex index, Probe& probe) {
                const float seconds_since_update = static_cast<float>(frame_count - probe.last_update_frame) / 60.f;
                const auto update_score = log(seconds_since_update);  // unqualified, as in the reference (:564): the overload the platform's headers pick
                const float num = distribution(rng);
                if (num < update_score * normalized_priority) {
                    if (!request_probe_update(index)) return false;
                    probe.is_valid = true;
                    probe.last_update_frame = frame_count;
                }
                return true;
            });
        }
    }

private:
    template <class F> static void foreach (Cascade& c, F func) {
        for (uint32_t x = 0; x < cascade_size_xz; x++)
            for (uint32_t y = 0; y < cascade_size_y; y++)
                for (uint32_t z = 0; z < cascade_size_xz; z++)
                    if (!func(ProbeIndex{x, y, z}, c.at(x, y, z))) return;
    }
    uint32_t update_budget;
    std::array<Cascade, num_cascades> cascades{};
    std::vector<ProbeIndex> probes_to_update;
};

// RenderCore/render/gi/irradiance_cache.{hpp,cpp}: atlases :94-183, cascade placement :298-372, copy_probes_to_new_texture
// :455-486, dispatch_probe_updates :585-724 (probe_tracing rays -> sah_probe_trace, then the update passes), overlay :203-245.
// When the scene has no acceleration structure yet, set_trace_results() is the seam for trace results produced elsewhere.
class IrradianceCache : public IGlobalIlluminator {
public:
    explicit IrradianceCache(RenderBackend& backend) : allocator(&backend.get_global_allocator()) {
        auto& alloc = backend.get_global_allocator();
        const uint32_t r11 = SAH_FORMAT_B10G11R11_UFLOAT_PACK32;
        for (int s = 0; s < 2; s++) {  // a / b sets, swapped by copy_probes_to_new_texture
            const std::string tag = s ? "_b" : "_a";
            set[s].rtgi = alloc.create_texture("probe_rtgi" + tag, r11, 32 * 7, 32 * 8, 32);
            set[s].light_cache = alloc.create_texture("probe_light_cache" + tag, r11, 32 * 13, 32 * 13, 32);
            set[s].depth = alloc.create_texture("probe_depth" + tag, SAH_FORMAT_R16G16_SFLOAT, 32 * 12, 32 * 12, 32);
            set[s].average = alloc.create_texture("probe_average" + tag, r11, 32, 32, 32);
            set[s].validity = alloc.create_texture("probe_validity" + tag, SAH_FORMAT_R8_UNORM, 32, 32, 32);
        }
        for (int c = 0; c < 4; c++) cascades[c].probe_spacing = 0.5f * std::pow(2.f, (float)c);
    }
    // irradiance_cache.cpp:298-372 places the cascades around the camera on a probe_spacing grid and records how many cells each
    // one moved; here the caller provides both (placement is scene logic, outside the hot path)
    void set_cascade(uint32_t c, const Vec3& min, float spacing, const Vec3& movement_cells) {
        for (int i = 0; i < 3; i++) { cascades[c].min[i] = min[i]; movement[c][i] = movement_cells[i]; }
        cascades[c].probe_spacing = spacing;
    }
    // irradiance_cache.cpp:496-583: which probes this frame's budget goes to.  The ray tracer (outside the path) then fills one
    // 20 x 20 layer of trace results per entry of get_probes_to_update(), in list order, and hands both to set_trace_results().
    ProbeScheduler& get_scheduler() { return scheduler; }
    void find_probes_to_update(uint32_t frame_count) { scheduler.find_probes_to_update(frame_count); }
    void set_trace_results(TextureHandle trace_results_in, const uint32_t* probes_to_update_device, uint32_t num_probes_in) {
        trace_results = trace_results_in;
        probes_to_update = probes_to_update_device;
        num_probes = num_probes_in;
    }
    void pre_render(RenderGraph& graph, const SceneView&, const RenderScene& scene, TextureHandle noise_tex) override {
        copy_probes_to_new_texture(graph);
        dispatch_probe_updates(graph, scene, noise_tex);
    }
    // irradiance_cache.cpp:585-650: upload the scheduler's list, 400 rays per probe into the trace results, then the update passes
    void dispatch_probe_updates(RenderGraph& graph, const RenderScene& scene, TextureHandle noise_tex) {
        const auto& list = scheduler.get_probes_to_update();
        if (!list.empty() && noise_tex && scene.get_raytracing_scene().is_built() && scene.sky.transmittance_lut && scene.sky.sky_view_lut) {
            if (list.size() > probe_list_capacity) {  // probes_to_update_buffer / trace_results_texture, sized for the scheduler's budget
                if (probe_list_device) (void)hipFree(probe_list_device);
                probe_list_capacity = std::max<size_t>(list.size(), 1024);
                if (hipMalloc(&probe_list_device, probe_list_capacity * 3 * sizeof(uint32_t)) != hipSuccess) throw std::runtime_error("Could not create probes_to_update_buffer");
                own_trace_results = allocator->create_texture("probe_trace_results", SAH_FORMAT_R16G16B16A16_SFLOAT, 20, 20, (uint32_t)probe_list_capacity);
            }
            static_assert(sizeof(ProbeScheduler::ProbeIndex) == 3 * sizeof(uint32_t), "uint3 list");
            if (hipMemcpy(probe_list_device, list.data(), list.size() * 3 * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) throw std::runtime_error("probe list upload failed");
            set_trace_results(own_trace_results, static_cast<const uint32_t*>(probe_list_device), (uint32_t)list.size());
            graph.add_pass(hip_pass("probe_tracing", [this, &scene, noise_tex](sah_ctx* ctx) {
                                sah_probe_trace_desc d{};
                                for (int c = 0; c < 4; c++) d.cascades[c] = cascades[c];
                                d.probes_to_update = probes_to_update;
                                d.num_probes = num_probes;
                                d.sun = &scene.sun.get_constants();
                                const sah_sky_luts sky{scene.sky.transmittance_lut->plane(), scene.sky.sky_view_lut->plane()};
                                const sah_plane noise = noise_tex->plane();
                                d.sky = &sky;
                                d.noise = &noise;
                                d.probe_irradiance = set[0].rtgi->desc;
                                d.probe_depth = set[0].depth->desc;
                                d.probe_validity = set[0].validity->desc;
                                d.probe_size[0] = 5;
                                d.probe_size[1] = 6;
                                d.trace_results = trace_results->desc;
                                return sah_probe_trace(ctx, &d);
                            }));
            scheduler.clear_probes_to_update();  // :722
        }
        dispatch_probe_updates(graph);
    }
    ~IrradianceCache() override {
        if (probe_list_device) (void)hipFree(probe_list_device);
    }
    void post_render(RenderGraph&, const SceneView&, const RenderScene&, const GBuffer&, TextureHandle) override {}
    void copy_probes_to_new_texture(RenderGraph& graph) {
        graph.add_pass(hip_pass("cascade_copy", [this](sah_ctx* ctx) {
                            const sah_probe_atlases a = atlases(0), b = atlases(1);
                            return sah_probe_copy(ctx, &a, &b, movement);
                        }));
        std::swap(set[0], set[1]);  // swap_probe_textures(): "a" is the current set again
    }
    void dispatch_probe_updates(RenderGraph& graph) {
        if (num_probes == 0 || trace_results == nullptr) return;  // irradiance_cache.cpp:590-592
        graph.add_pass(hip_pass("probe updates", [this](sah_ctx* ctx) {
                            const sah_probe_atlases a = atlases(0);
                            return sah_probe_update(ctx, &a, &trace_results->desc, probes_to_update, num_probes);
                        }));
    }
    void get_lighting_resource_usages(TextureUsageList& textures, BufferUsageList&) const override {
        for (TextureHandle t : {set[0].rtgi, set[0].depth, set[0].validity}) textures.push_back({t, kStageFragmentShader, kAccessShaderRead, kLayoutShaderReadOnly});
    }
    void draw_debug_overlays(RenderGraph&, const SceneView&, const GBuffer&, TextureHandle) override {}  // probe spheres (irradiance_cache.cpp:308-349): debug view
    void render_to_lit_scene(CommandBuffer& commands, BufferHandle, TextureHandle, TextureHandle) const override {  // add_to_lit_scene, :287-306
        sah_gi& gi = commands.lighting.gi;
        gi = sah_gi{};
        commands.lighting.has_gi = true;
        gi.kind = SAH_GI_CACHE;
        gi.probe_irradiance = set[0].rtgi->desc;
        gi.probe_depth = set[0].depth->desc;
        gi.probe_validity = set[0].validity->desc;
        for (int c = 0; c < 4; c++) gi.probe_cascades[c] = cascades[c];
        gi.probe_size[0] = 5;
        gi.probe_size[1] = 6;
        gi.cache_debug_mode = 0;
    }
    sah_probe_atlases atlases(int s) const {
        return sah_probe_atlases{set[s].rtgi->desc, set[s].light_cache->desc, set[s].depth->desc, set[s].average->desc, set[s].validity->desc};
    }

private:
    struct Set {
        TextureHandle rtgi{}, light_cache{}, depth{}, average{}, validity{};
    } set[2];
    std::array<sah_probe_cascade, 4> cascades{};
    float movement[4][3] = {};
    TextureHandle trace_results{};
    const uint32_t* probes_to_update = nullptr;
    ProbeScheduler scheduler;
    uint32_t num_probes = 0;
    ResourceAllocator* allocator = nullptr;
    void* probe_list_device = nullptr;
    size_t probe_list_capacity = 0;
    TextureHandle own_trace_results{};
};

// RenderCore/render/gi/rtgi.{hpp,cpp}: post_render (:69-139) traces one GI ray per pixel (sah_rtgi_trace) into the two per-pixel
// textures — ray direction + distance, ray irradiance — that the reconstruction overlay (rtgi.cpp:160-188,
// gi/rtgi/overlay.frag.slang:68-117) binds in the NEXT frame's Lighting pass.  It owns an IrradianceCache, as the reference does
// (rtgi.hpp: irradiance_cache).  set_ray_textures() remains for ray textures produced elsewhere.
class RayTracedGlobalIllumination : public IGlobalIlluminator {
public:
    explicit RayTracedGlobalIllumination(RenderBackend& backend) : cache(backend), allocator(&backend.get_global_allocator()) {}
    void set_ray_textures(TextureHandle ray_texture_in, TextureHandle ray_irradiance_in) {
        ray_texture = ray_texture_in;
        ray_irradiance = ray_irradiance_in;
    }
    void set_reconstruction(uint32_t num_samples, float size) {  // r.GI.Reconstruction.NumSamples / .Size
        num_extra_rays = num_samples;
        extra_ray_radius = size;
    }
    // r.GI.NumBounces (rtgi.cpp:10).  The reference pushes it to the generator (rtgi.cpp:131), which never reads it and sets
    // payload.remaining_bounces = 0 (rtgi.rt.slang:88): 0 here is the reference's image, 1 or 2 what its hit stage (gltf_basic_pbr.slang:481-517)
    // computes once the generator forwards the constant
    void set_forwarded_bounces(uint32_t n) { forwarded_bounces = n; }
    IrradianceCache& get_irradiance_cache() { return cache; }
    TextureHandle get_ray_texture() const { return ray_texture; }
    TextureHandle get_ray_irradiance() const { return ray_irradiance; }
    void pre_render(RenderGraph& graph, const SceneView& view, const RenderScene& scene, TextureHandle noise_tex) override {
        cache.pre_render(graph, view, scene, noise_tex);  // rtgi.cpp:60-75: the cache updates first
    }
    void post_render(RenderGraph& graph, const SceneView& view, const RenderScene& scene, const GBuffer& gbuffer, TextureHandle noise_tex) override {
        if (!scene.get_raytracing_scene().is_built() || !noise_tex || !scene.sky.transmittance_lut || !scene.sky.sky_view_lut) return;
        const uint32_t w = gbuffer.depth->desc.width, h = gbuffer.depth->desc.height;
        if (ray_texture == nullptr || ray_texture->desc.width != w || ray_texture->desc.height != h)
            ray_texture = allocator->create_texture("rtgi_params", SAH_FORMAT_R16G16B16A16_SFLOAT, w, h);
        if (ray_irradiance == nullptr || ray_irradiance->desc.width != w || ray_irradiance->desc.height != h)
            ray_irradiance = allocator->create_texture("rtgi_irradiance", SAH_FORMAT_R16G16B16A16_SFLOAT, w, h);
        graph.add_pass(hip_pass("ray_traced_global_illumination", [this, &view, &scene, gbuffer, noise_tex](sah_ctx* ctx) {
                            const sah_sky_luts sky{scene.sky.transmittance_lut->plane(), scene.sky.sky_view_lut->plane()};
                            const sah_plane depth = gbuffer.depth->plane(), normals = gbuffer.normals->plane(), noise = noise_tex->plane();
                            const sah_plane rb = ray_texture->plane(), ri = ray_irradiance->plane();
                            if (int rc = sah_rt_set_bounces(ctx, forwarded_bounces); rc != SAH_OK) return rc;
                            const int rc = sah_rtgi_trace(ctx, &view.get_gpu_data(), &scene.sun.get_constants(), &sky, &depth, &normals, &noise, &rb, &ri);
                            sah_rt_set_bounces(ctx, 0);  // (the context's setting also governs the irradiance cache's probe rays)
                            return rc;
                        }));
    }
    void get_lighting_resource_usages(TextureUsageList& textures, BufferUsageList& buffers) const override {
        cache.get_lighting_resource_usages(textures, buffers);
        for (TextureHandle t : {ray_texture, ray_irradiance}) textures.push_back({t, kStageFragmentShader, kAccessShaderRead, kLayoutShaderReadOnly});
    }
    void draw_debug_overlays(RenderGraph&, const SceneView&, const GBuffer&, TextureHandle) override {}
    void render_to_lit_scene(CommandBuffer& commands, BufferHandle, TextureHandle, TextureHandle noise_tex) const override {
        sah_gi& gi = commands.lighting.gi;
        gi = sah_gi{};
        commands.lighting.has_gi = true;
        gi.kind = SAH_GI_RTGI;
        gi.ray_buffer = ray_texture->plane();
        gi.ray_irradiance = ray_irradiance->plane();
        if (noise_tex) gi.noise = noise_tex->plane();
        gi.num_extra_rays = noise_tex ? num_extra_rays : 0;
        gi.extra_ray_radius = extra_ray_radius;
    }

private:
    IrradianceCache cache;
    ResourceAllocator* allocator = nullptr;
    TextureHandle ray_texture{}, ray_irradiance{};
    uint32_t forwarded_bounces = 0;
    uint32_t num_extra_rays = 0;
    float extra_ray_radius = 16.f;
};

// RenderCore/render/phase/ambient_occlusion_phase.cpp:157-189: r.AO.Mode = Off clears the target to 1.0 (:167-179), RTAO traces one
// occlusion ray per pixel (evaluate_rtao, :357-397); CACAO is a vendor SDK and hands its result over as the AO plane.
enum class AoTechnique { Off = 0, CACAO = 1, RTAO = 2 };
class AmbientOcclusionPhase {
public:
    AoTechnique technique = AoTechnique::RTAO;  // r.AO (ambient_occlusion_phase.cpp:15-17: RTAO is the reference's default)
    uint32_t rtao_samples = 1;                  // r.AO.RTAO.SamplesPerPixel (:19-21)
    float ao_radius = 8.0f;                     // r.AO.MaxRayDistance (:23-25)
    void generate_ao(RenderGraph& graph, TextureHandle ao_out) {  // the Off path (no scene needed)
        graph.add_pass(hip_pass("Clear AO", [ao_out](sah_ctx* ctx) {
                            const sah_plane p = ao_out->plane();
                            return sah_ao_clear(ctx, &p);
                        }));
    }
    // ambient_occlusion_phase.hpp: generate_ao(graph, view, scene, noise, gbuffer_normals, gbuffer_depth, ao_out)
    void generate_ao(RenderGraph& graph, const SceneView& view, const RenderScene& scene, const NoiseTexture& noise, TextureHandle gbuffer_normals,
                     TextureHandle gbuffer_depth, TextureHandle ao_out) {
        if (technique != AoTechnique::RTAO || !scene.get_raytracing_scene().is_built()) return generate_ao(graph, ao_out);
        TextureHandle layer = noise.get_layer(frame_index);
        frame_index++;
        graph.add_pass(hip_pass("Ray traced ambient occlusion", [this, &view, layer, gbuffer_normals, gbuffer_depth, ao_out](sah_ctx* ctx) {
                            const sah_plane d = gbuffer_depth->plane(), n = gbuffer_normals->plane(), z = layer->plane(), o = ao_out->plane();
                            return sah_rtao(ctx, &view.get_gpu_data(), &d, &n, &z, rtao_samples, ao_radius, &o);
                        }));
    }

private:
    uint32_t frame_index = 0;
};

// ---- LightingPhase (RenderCore/render/phase/lighting_phase.hpp:17-57, .cpp:34-134) -------------------------------------
// RenderCore/render/phase/gbuffer_phase.cpp:17-97 (and the depth pre-pass it relies on, phase/depth_culling_phase.cpp): visibility and
// the four G-buffer targets in one compute pass.  The indirect draw buffers of the reference (GPU culling results) have no
// counterpart: every primitive of the scene is submitted.
class GbufferPhase {
public:
    void render(RenderGraph& graph, const RenderScene& scene, const GBuffer& gbuffer, const SceneView& player_view) {
        graph.add_pass(hip_pass("gbuffer", [&scene, &gbuffer, &player_view](sah_ctx* ctx) {
                            const sah_gbuffer g = {gbuffer.color->plane(), gbuffer.normals->plane(), gbuffer.data->plane(), gbuffer.emission->plane(),
                                                   gbuffer.depth->plane()};
                            return sah_gbuffer_render(ctx, &scene.geometry, &player_view.get_gpu_data(), &g, nullptr);
                        }));
    }
};

class LightingPhase {
public:
    void set_scene(RenderScene& scene_in) { scene = &scene_in; }
    // lighting_phase.hpp:33-43 / lighting_phase.cpp:34-134, same parameters.  The "Lighting" render pass clears lit_scene and runs, in
    // this order, the CSM-mode sun, the GI overlay, the emissive pass and the sky; the RT-mode sun follows as its own dispatch.  Each of
    // them records into the command buffer; the fused sah_lighting call (which keeps that order and its fp16 blend roundings) is
    // submitted at the end.
    void render(RenderGraph& render_graph, const SceneView& view, const GBuffer& gbuffer, TextureHandle lit_scene_texture, TextureHandle ao_texture,
                const IGlobalIlluminator* gi, std::optional<TextureHandle> vrsaa_shading_rate_image, const NoiseTexture& noise, TextureHandle noise_2d) {
        (void)vrsaa_shading_rate_image;  // VRSAA (sampling_rate_calculator.cpp) is AA plumbing outside the path: full rate
        if (scene == nullptr) return;     // silent no-op, lighting_phase.cpp:47-49
        auto& sun = scene->get_sun_light();
        TextureUsageList texture_usages;
        BufferUsageList buffer_usages;
        for (TextureHandle t : {gbuffer.color, gbuffer.normals, gbuffer.data, gbuffer.emission, gbuffer.depth})
            texture_usages.push_back({t, kStageFragmentShader, kAccessShaderRead, kLayoutShaderReadOnly});
        if (gi) gi->get_lighting_resource_usages(texture_usages, buffer_usages);
        DynamicRenderingPass pass;
        pass.name = "Lighting";
        pass.textures = texture_usages;
        pass.buffers = buffer_usages;
        pass.color_attachments = {RenderingAttachmentInfo{lit_scene_texture, /*VK_ATTACHMENT_LOAD_OP_CLEAR*/ 1}};
        pass.execute = [&, lit_scene_texture, ao_texture, gi, noise_2d](CommandBuffer& commands) {
            auto& rec = commands.lighting;
            rec = CommandBuffer::LightingRecording{};
            rec.open = true;
            rec.gbuffer = sah_gbuffer{gbuffer.color->plane(), gbuffer.normals->plane(), gbuffer.data->plane(), gbuffer.emission->plane(), gbuffer.depth->plane()};
            rec.lit = lit_scene_texture->plane();
            rec.view = static_cast<const sah_view_data*>(view.get_buffer()->data);
            rec.sun = &sun.get_constants();  // shadow_mode in the constants selects which sun term the fused pass evaluates
            if (sun.get_shadow_mode() == SunShadowMode::CascadedShadowMaps) sun.render(commands, view);
            if (gi) gi->render_to_lit_scene(commands, view.get_buffer(), ao_texture, noise_2d);
            add_emissive_lighting(commands);
            scene->get_sky().render_sky(commands, view.get_buffer(), sun.get_constant_buffer(), gbuffer.depth);
        };
        render_graph.add_render_pass(std::move(pass));
        if (sun.get_shadow_mode() == SunShadowMode::RayTracing) sun.raytrace(render_graph, view, gbuffer, *scene, lit_scene_texture, noise);
        ComputePass submit;
        submit.name = "Lighting (fused HIP dispatch)";
        submit.execute = [](CommandBuffer& commands) { commands.submit_lighting(); };
        render_graph.add_pass(std::move(submit));
    }

private:
    void add_emissive_lighting(CommandBuffer&) const {}  // lighting_phase.cpp:176-186: the emission plane is part of the bound G-buffer; nothing else to bind
    RenderScene* scene = nullptr;
};

// ---- Bloomer (RenderCore/render/bloomer.hpp:15-17, .cpp:38-285) -----------------------------------------------------------
class Bloomer {
public:
    explicit Bloomer(RenderBackend& backend_in) : backend(backend_in) {}
    void fill_bloom_tex(RenderGraph& graph, TextureHandle scene_color) {
        if (mips.empty()) create_bloom_tex(scene_color);
        graph.add_pass(hip_pass("Bloom", [this, scene_color](sah_ctx* ctx) {
                            const sah_plane s = scene_color->plane();
                            return sah_bloom(ctx, &s, &chain);
                        }));
    }
    // "Copy scene" (scene_renderer.cpp:502-527) directly followed by the bloom pass (bloomer.cpp:50-72), as the frame records them: the copy
    // and the first bloom dispatch in one pass over lit_scene, then the rest of the chain.  Same bits as evaluate_antialiasing_none() +
    // fill_bloom_tex(); extents the fused form does not take fall back to the two passes inside the library.
    void copy_scene_and_fill_bloom_tex(RenderGraph& graph, TextureHandle lit_scene, TextureHandle antialiased_scene) {
        if (mips.empty()) create_bloom_tex(antialiased_scene);
        graph.add_pass(hip_pass("Copy scene + Bloom", [this, lit_scene, antialiased_scene](sah_ctx* ctx) {
                            const sah_plane l = lit_scene->plane(), a = antialiased_scene->plane();
                            if (int rc = sah_copy_scene_bloom_mip0_rows(ctx, &l, &a, &chain, 0, 0, 0, 0); rc != SAH_OK) return rc;
                            return sah_bloom_from_mip0(ctx, &a, &chain);
                        }));
    }
    TextureHandle get_bloom_tex() const { return bloom_tex.get(); }  // bloomer.hpp:17

private:
    void create_bloom_tex(TextureHandle scene_color) {  // bloomer.cpp:268-285: scene/2, six mips, scene format
        uint32_t w = scene_color->desc.width / 2, h = scene_color->desc.height / 2;
        chain.num_mips = 6;  // r.bloom.NumMips
        for (uint32_t m = 0; m < chain.num_mips; m++) {
            mips.push_back(backend.get_global_allocator().create_texture("Bloom texture mip " + std::to_string(m), SAH_FORMAT_R16G16B16A16_SFLOAT,
                                                                          w ? w : 1, h ? h : 1));
            chain.mips[m] = mips.back()->plane();
            w = w / 2 ? w / 2 : 1;
            h = h / 2 ? h / 2 : 1;
        }
        bloom_tex = std::make_unique<Texture>();  // one handle for the whole chain, as the reference's mipped "Bloom texture"
        bloom_tex->name = "Bloom texture";
        bloom_tex->desc = mips[0]->desc;
        bloom_tex->mips = chain;
    }
    RenderBackend& backend;
    std::vector<TextureHandle> mips;
    sah_mipchain chain{};
    std::unique_ptr<Texture> bloom_tex;
};

// ---- UiPhase::draw_scene_image (RenderCore/render/phase/ui_phase.hpp:27, .cpp:98-113) -------------------------------------
class UiPhase {
public:
    void set_resources(TextureHandle scene_color_in, TextureHandle swapchain_in) { scene_color = scene_color_in; swapchain = swapchain_in; }
    // ui_phase.hpp:27: called from inside the "UI" render pass (scene_renderer.cpp:426-449); draw_scene_image (:98-113) is the part on the path
    void render(CommandBuffer& commands, const SceneView&, TextureHandle bloom_texture) const {
        const sah_plane s = scene_color->plane(), o = swapchain->plane();
        commands.check(sah_tonemap(commands.get_context(), &s, &bloom_texture->mips, &o, 0, 0), "UI");
    }

private:
    TextureHandle scene_color = nullptr, swapchain = nullptr;
};

// "Copy scene" (AA = None), RenderCore/render/scene_renderer.cpp:502-527
inline void evaluate_antialiasing_none(RenderGraph& graph, TextureHandle lit_scene, TextureHandle antialiased_scene) {
    graph.add_pass(hip_pass("Copy scene", [lit_scene, antialiased_scene](sah_ctx* ctx) {
                        const sah_plane a = lit_scene->plane(), b = antialiased_scene->plane();
                        return sah_copy_scene(ctx, &a, &b);
                    }));
}

// ---- the row-sharded frame (no reference counterpart: north_star's "frames shard by screen-tile rows across the 8 GPUs") ------------------
// Which rows a rank shades, copies, reduces and composites so that the two exchanges of sah_chain_submit — the quarter-resolution bloom
// mip 1, the final R8G8B8A8 image — reassemble exactly what one GPU computes.  Dependencies, in rows (include/sah_hip.h; the same integer
// arithmetic as androidrenderer_amd/shard.py: chain_plan, and tests/test_shard_chain.py holds the two against each other):
//   final image row y     samples the scene upside down (scene_upsample.frag, fullscreen.vert: v = 1 - (y + 0.5) / H): antialiased rows
//                         H - 1 - y +- 1, bloom mip 0 rows within 3 of (1 - (y + 0.5) / H) * H0 - 0.5, every smaller mip (global);
//   antialiased row j     "Copy scene": lit rows j - 1 .. j + 1, and its sampler REPEATS (row 0 taps row H - 1 and the other way round);
//   bloom mip m row j     the rows of its source that sah_bloom_source_rows names.
// Rank r owns mip 1 rows [r q, (r + 1) q) and the final rows of slot N - 1 - r (the vertical flip: those are the rows whose scene rows it has).
struct ShardPlan {
    sah_chain_plan chain{};          // what sah_chain_create takes (allocations: mip1_rows_per_rank * world and rows_per_rank * world rows)
    uint32_t lit_rows[2]{};          // rows this rank's sah_lighting shades (sah_lighting_desc::row_begin / row_end) ...
    uint32_t lit_wrap_rows[2]{};     // ... plus the row on the opposite edge that the REPEAT sampler taps ((0, 0): none): a second sah_lighting call
    uint32_t mip0_height = 0, mip1_height = 0;
};
inline ShardPlan shard_chain_plan(uint32_t height, uint32_t world, uint32_t rank) {
    auto clip = [](int64_t a, int64_t b, int64_t n, uint32_t out[2]) {
        a = std::max<int64_t>(0, std::min(a, n));
        b = std::max<int64_t>(0, std::min(b, n));
        out[0] = (uint32_t)a;
        out[1] = (uint32_t)std::max(a, b);
    };
    auto floordiv = [](int64_t a, int64_t b) { return a >= 0 ? a / b : -((-a + b - 1) / b); };
    auto hull = [](bool& any, int64_t& lo, int64_t& hi, int64_t a, int64_t b) {
        if (b <= a) return;
        lo = any ? std::min(lo, a) : a;
        hi = any ? std::max(hi, b) : b;
        any = true;
    };
    ShardPlan p;
    const int64_t H = height, N = world, per = (H + N - 1) / N;
    const int64_t h0 = std::max<int64_t>(1, H / 2), h1 = std::max<int64_t>(1, h0 / 2), q = (h1 + N - 1) / N, slot = N - 1 - (int64_t)rank;
    p.mip0_height = (uint32_t)h0;
    p.mip1_height = (uint32_t)h1;
    p.chain.rows_per_rank = (uint32_t)per;
    p.chain.out_allocated_rows = (uint32_t)(per * N);
    p.chain.mip1_rows_per_rank = (uint32_t)q;
    p.chain.mip1_allocated_rows = (uint32_t)(q * N);
    clip(slot * per, (slot + 1) * per, H, p.chain.out_rows);
    clip((int64_t)rank * q, ((int64_t)rank + 1) * q, h1, p.chain.mip1_rows);
    bool any0 = false, any_aa = false;
    int64_t lo0 = 0, hi0 = 0, lo_aa = 0, hi_aa = 0;
    const uint32_t* out = p.chain.out_rows;
    if (out[1] > out[0]) {
        hull(any_aa, lo_aa, hi_aa, H - out[1] - 1, H - out[0] + 1);
        const int64_t slack = (H == 2 * h0) ? 0 : 1;
        const int64_t p_lo = floordiv((2 * (H - ((int64_t)out[1] - 1)) - 1) * h0 - H, 2 * H), p_hi = floordiv((2 * (H - (int64_t)out[0]) - 1) * h0 - H, 2 * H);
        hull(any0, lo0, hi0, p_lo - 1 - slack, p_hi + 2 + slack + 1);
    }
    if (p.chain.mip1_rows[1] > p.chain.mip1_rows[0]) {
        // (the unclipped window: the plan takes the hull of its needs before it clips; sah_bloom_source_rows returns the clipped one)
        const int64_t j0 = p.chain.mip1_rows[0], j1 = p.chain.mip1_rows[1], s1 = (h0 == 2 * h1) ? 0 : 1;
        hull(any0, lo0, hi0, floordiv((2 * j0 + 1) * h0 - h1 - 4 * h1, 2 * h1) - s1, floordiv((2 * (j1 - 1) + 1) * h0 - h1 + 4 * h1, 2 * h1) + 1 + s1 + 1);
    }
    if (any0) clip(lo0, hi0, h0, p.chain.mip0_rows);
    if (p.chain.mip0_rows[1] > p.chain.mip0_rows[0]) {
        const int64_t j0 = p.chain.mip0_rows[0], j1 = p.chain.mip0_rows[1], s0 = (H == 2 * h0) ? 0 : 1;
        hull(any_aa, lo_aa, hi_aa, floordiv((2 * j0 + 1) * H - h0 - 4 * h0, 2 * h0) - s0, floordiv((2 * (j1 - 1) + 1) * H - h0 + 4 * h0, 2 * h0) + 1 + s0 + 1);
    }
    if (any_aa) {
        clip(lo_aa, hi_aa, H, p.chain.aa_rows);
        clip((int64_t)p.chain.aa_rows[0] - 1, (int64_t)p.chain.aa_rows[1] + 1, H, p.lit_rows);
        if (p.chain.aa_rows[0] == 0 && p.lit_rows[1] < height) {
            p.lit_wrap_rows[0] = height - 1;
            p.lit_wrap_rows[1] = height;
        } else if (p.chain.aa_rows[1] == height && p.lit_rows[0] > 0) {
            p.lit_wrap_rows[0] = 0;
            p.lit_wrap_rows[1] = 1;
        }
    }
    return p;
}

}  // namespace sah
