// The two producer passes through the C++ host façade (include/sah_host.hpp): DirectionalLight::update_shadow_cascades +
// render_shadows (RenderCore/render/directional_light.cpp:84-230,286-327) and GbufferPhase::render
// (RenderCore/render/phase/gbuffer_phase.cpp:17-97).  The mesh comes from a file written by tests/test_host_facade_gpu.py; the
// uniform blocks built here go back with the images so that the test can hand the very same blocks to the oracle.
//
//   host_raster <in.bin> <out.bin>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "sah_host.hpp"

static std::vector<unsigned char> read_blob(FILE* f, size_t n) {
    std::vector<unsigned char> v(n);
    if (n && fread(v.data(), 1, n, f) != n) { fprintf(stderr, "short read\n"); exit(2); }
    return v;
}
static void* to_device(const std::vector<unsigned char>& v) {
    void* p = nullptr;
    if (v.empty()) return nullptr;
    if (hipMalloc(&p, v.size()) != hipSuccess || hipMemcpy(p, v.data(), v.size(), hipMemcpyHostToDevice) != hipSuccess) exit(3);
    return p;
}

int main(int argc, char** argv) {
    if (argc != 3) { fprintf(stderr, "usage: host_raster in.bin out.bin\n"); return 2; }
    FILE* in = fopen(argv[1], "rb");
    if (!in) { perror("open input"); return 2; }
    uint32_t hdr[8];  // W, H, shadow resolution, vertices, indices, primitives, materials, textures
    if (fread(hdr, 4, 8, in) != 8) return 2;
    const uint32_t W = hdr[0], H = hdr[1], R = hdr[2];

    using namespace sah;
    RenderBackend backend(0);
    auto& alloc = backend.get_global_allocator();

    RenderScene scene;
    scene.geometry.num_vertices = hdr[3];
    scene.geometry.num_indices = hdr[4];
    scene.geometry.num_primitives = hdr[5];
    scene.geometry.num_materials = hdr[6];
    scene.geometry.vertex_positions = (const float*)to_device(read_blob(in, (size_t)hdr[3] * 12));
    scene.geometry.vertex_data = (const sah_vertex_data*)to_device(read_blob(in, (size_t)hdr[3] * sizeof(sah_vertex_data)));
    scene.geometry.indices = (const uint32_t*)to_device(read_blob(in, (size_t)hdr[4] * 4));
    scene.geometry.primitives = (const sah_primitive*)to_device(read_blob(in, (size_t)hdr[5] * sizeof(sah_primitive)));
    scene.geometry.materials = (const sah_material*)to_device(read_blob(in, (size_t)hdr[6] * sizeof(sah_material)));
    // bindless textures (texture_descriptor_pool.hpp): per texture {format, levels, sampler (10 words)}, then per level {w, h} + texels;
    // then one sah_material_textures per material
    TextureDescriptorPool texture_pool(backend);
    for (uint32_t t = 0; t < hdr[7]; t++) {
        uint32_t th[12];
        if (fread(th, 4, 12, in) != 12) return 2;
        sah_sampler smp;
        static_assert(sizeof(smp) == 40, "sampler words");
        memcpy(&smp, th + 2, 40);
        std::vector<TextureHandle> levels;
        for (uint32_t l = 0; l < th[1]; l++) {
            uint32_t wh[2];
            if (fread(wh, 4, 2, in) != 2) return 2;
            TextureHandle level = alloc.create_texture("material texture " + std::to_string(t) + " mip " + std::to_string(l), th[0], wh[0], wh[1]);
            const auto texels = read_blob(in, (size_t)wh[0] * wh[1] * 4);
            alloc.upload(level, texels.data(), wh[0] * 4);
            levels.push_back(level);
        }
        if (texture_pool.create_texture_srv(levels, smp) != t) return 4;
    }
    // one 128 x 128 blue-noise layer (RenderCore/render/noise_texture.cpp; the PNGs are not in the reference tree: the test sends noise)
    NoiseTexture stbn_3d_unitvec;
    {
        TextureHandle layer = alloc.create_texture("stbn_unitvec3_2Dx1D_128x128x64_0", SAH_FORMAT_R8G8B8A8_UNORM, 128, 128);
        const auto texels = read_blob(in, 128 * 128 * 4);
        alloc.upload(layer, texels.data(), 128 * 4);
        stbn_3d_unitvec.layers.push_back(layer);
        stbn_3d_unitvec.resolution[0] = stbn_3d_unitvec.resolution[1] = 128;
        stbn_3d_unitvec.num_layers = 1;
    }
    if (hdr[7]) {
        texture_pool.commit_descriptors();  // "should be called at start of frame"
        scene.geometry.textures = texture_pool.get_descriptor_set();
        scene.geometry.num_textures = texture_pool.size();
        scene.geometry.material_textures = (const sah_material_textures*)to_device(read_blob(in, (size_t)hdr[6] * sizeof(sah_material_textures)));
    }
    fclose(in);

    SceneView view;  // start-up camera of the reference: scene_renderer.cpp:53-54,105-116
    view.rotate(0.f, 90.f * 3.14159265358979f / 180.f);
    view.set_position({-7.f, 1.f, 0.f});
    view.set_render_resolution(W, H);
    view.set_perspective_projection(75.f, (float)W / (float)H, 0.05f);
    view.update_transforms();

    scene.sun.set_shadow_mode(SunShadowMode::CascadedShadowMaps);
    scene.sun.shadowmap_handle = alloc.create_texture("Sun shadowmap", SAH_FORMAT_D16_UNORM, R, R, 4);
    scene.sun.update_shadow_cascades(view, 4, 128.f, 0.95f, R);

    GBuffer gbuffer;
    gbuffer.color = alloc.create_texture("gbuffer_color", SAH_FORMAT_R8G8B8A8_SRGB, W, H);
    gbuffer.normals = alloc.create_texture("gbuffer_normals", SAH_FORMAT_R16G16B16A16_SFLOAT, W, H);
    gbuffer.data = alloc.create_texture("gbuffer_data", SAH_FORMAT_R8G8B8A8_UNORM, W, H);
    gbuffer.emission = alloc.create_texture("gbuffer_emission", SAH_FORMAT_R8G8B8A8_SRGB, W, H);
    gbuffer.depth = alloc.create_texture("gbuffer_depth", SAH_FORMAT_D32_SFLOAT, W, H);

    RenderGraph graph{backend};
    scene.sun.render_shadows(graph, scene.geometry);
    GbufferPhase gbuffer_phase;
    gbuffer_phase.render(graph, scene, gbuffer, view);
    // the LPV's own producers: clear, RSM, VPL extraction and injection (light_propagation_volume.cpp:548-697)
    LightPropagationVolume lpv(backend, 4, 4);
    lpv.update_cascade_transforms(view, scene.sun);
    lpv.pre_render(graph, view, scene, nullptr);
    lpv.inject_indirect_sun_light(graph, scene);
    // ray tracing (scene_renderer.cpp:247-251, 383-401): TLAS over the scene's primitives, RTAO, and the shadow rays of the RT-mode sun
    for (uint32_t p = 0; p < scene.geometry.num_primitives; p++) scene.get_raytracing_scene().add_primitive(p);
    scene.get_raytracing_scene().finalize(graph);
    TextureHandle ao = alloc.create_texture("ao", SAH_FORMAT_R32_SFLOAT, W, H);
    AmbientOcclusionPhase ao_phase;
    ao_phase.technique = AoTechnique::RTAO;
    ao_phase.generate_ao(graph, view, scene, stbn_3d_unitvec, gbuffer.normals, gbuffer.depth, ao);
    scene.sun.shadow_mask = alloc.create_texture("sun shadow mask", SAH_FORMAT_R32_SFLOAT, W, H);
    TextureHandle lit_scene = alloc.create_texture("lit_scene", SAH_FORMAT_R16G16B16A16_SFLOAT, W, H);
    scene.sun.get_constants().num_shadow_samples = 2.0f;
    scene.sun.raytrace(graph, view, gbuffer, scene, lit_scene, stbn_3d_unitvec);
    // the GI rays (rtgi.cpp:41-139): sky LUTs for the miss stage, the irradiance cache traces and folds the scheduler's probes
    // (irradiance_cache.cpp:585-724), then one GI ray per pixel
    scene.sky.create_luts(alloc);
    scene.sky.update_sky_luts(graph, Vec3{0.3f, 0.8f, 0.2f});
    RayTracedGlobalIllumination rtgi(backend);
    IrradianceCache& cache = rtgi.get_irradiance_cache();
    for (uint32_t c = 0; c < 4; c++) {
        const float spacing = 0.5f * (float)(1u << c);
        cache.set_cascade(c, Vec3{-16.f * spacing + 0.013f, 1.f - 4.f * spacing, -16.f * spacing + 0.021f}, spacing, Vec3{0.f, 0.f, 0.f});
    }
    const ProbeScheduler::ProbeIndex wanted[6] = {{16, 3, 16}, {15, 2, 17}, {10, 9, 20}, {18, 12, 14}, {16, 19, 16}, {16, 27, 15}};
    for (const auto& id : wanted) cache.get_scheduler().request_probe_update(id);
    rtgi.pre_render(graph, view, scene, stbn_3d_unitvec.get_layer(0));
    rtgi.post_render(graph, view, scene, gbuffer, stbn_3d_unitvec.get_layer(0));
    graph.finish();
    for (const auto& e : graph.get_errors()) fprintf(stderr, "pass failed: %s\n", e.c_str());
    if (!graph.get_errors().empty()) return 1;

    FILE* out = fopen(argv[2], "wb");
    if (!out) { perror("open output"); return 2; }
    fwrite(&view.get_gpu_data(), sizeof(sah_view_data), 1, out);
    fwrite(&scene.sun.get_constants(), sizeof(sah_sun_light_constants), 1, out);
    std::vector<unsigned char> buf((size_t)R * R * 4 * 2);
    alloc.download(scene.sun.shadowmap_handle, buf.data(), R * 2);
    fwrite(buf.data(), 1, buf.size(), out);
    const struct { TextureHandle t; uint32_t bpp; } planes[5] = {{gbuffer.color, 4}, {gbuffer.normals, 8}, {gbuffer.data, 4}, {gbuffer.emission, 4}, {gbuffer.depth, 4}};
    for (const auto& p : planes) {
        buf.resize((size_t)W * H * p.bpp);
        alloc.download(p.t, buf.data(), W * p.bpp);
        fwrite(buf.data(), 1, buf.size(), out);
    }
    fwrite(lpv.get_cascade_matrices(), sizeof(sah_lpv_cascade_matrices), 4, out);
    for (int c = 0; c < 3; c++) {  // injected A volumes
        std::vector<unsigned char> v((size_t)128 * 32 * 32 * 8);
        alloc.download(lpv.get_volume(c), v.data(), 128 * 8);
        fwrite(v.data(), 1, v.size(), out);
    }
    for (TextureHandle t : {ao, scene.sun.shadow_mask}) {
        buf.resize((size_t)W * H * 4);
        alloc.download(t, buf.data(), W * 4);
        fwrite(buf.data(), 1, buf.size(), out);
    }
    for (TextureHandle t : {scene.sky.transmittance_lut, scene.sky.sky_view_lut, rtgi.get_ray_texture(), rtgi.get_ray_irradiance()}) {
        buf.resize((size_t)t->desc.width * t->desc.height * 8);
        alloc.download(t, buf.data(), t->desc.width * 8);
        fwrite(buf.data(), 1, buf.size(), out);
    }
    const sah_probe_atlases atl = cache.atlases(0);
    for (const sah_volume* v : {&atl.rtgi, &atl.light_cache, &atl.depth, &atl.average, &atl.validity}) {
        Texture t;
        t.desc = *v;
        const uint32_t row = v->width * format_bytes(v->format);
        buf.resize((size_t)row * v->height * v->depth);
        alloc.download(&t, buf.data(), row);
        fwrite(buf.data(), 1, buf.size(), out);
    }
    fclose(out);
    return 0;
}
