// Renders one frame through the C++ host façade (include/sah_host.hpp), the way SceneRenderer::render drives the
// reference's phases (RenderCore/render/scene_renderer.cpp:318-449): GI pre_render / post_render -> AO -> lighting -> copy scene ->
// bloom -> the "UI" render pass, with the reference's own call signatures (gi->post_render(...), lighting_pass.render(... nine
// arguments ...), ui_phase.render(commands, view, bloom)) and every RenderGraph verb the reference has.  Inputs come from a file written by tests/test_host_facade_gpu.py; the uniform blocks this program
// builds are written back with the images so that the test can feed the very same blocks to the oracle.
//
//   host_frame <in.bin> <out.bin>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "sah_host.hpp"

static std::vector<unsigned char> read_blob(FILE* f, size_t n) {
    std::vector<unsigned char> v(n);
    if (fread(v.data(), 1, n, f) != n) { fprintf(stderr, "short read\n"); exit(2); }
    return v;
}

int main(int argc, char** argv) {
    if (argc != 3) { fprintf(stderr, "usage: host_frame in.bin out.bin\n"); return 2; }
    FILE* in = fopen(argv[1], "rb");
    if (!in) { perror("open input"); return 2; }
    uint32_t hdr[4];
    if (fread(hdr, 4, 4, in) != 4) return 2;
    const uint32_t W = hdr[0], H = hdr[1], sun_mode = hdr[2], lpv_steps = hdr[3];

    using namespace sah;
    RenderBackend backend(0);
    auto& alloc = backend.get_global_allocator();

    GBuffer gbuffer;
    gbuffer.color = alloc.create_texture("gbuffer_color", SAH_FORMAT_R8G8B8A8_SRGB, W, H);
    gbuffer.normals = alloc.create_texture("gbuffer_normals", SAH_FORMAT_R16G16B16A16_SFLOAT, W, H);
    gbuffer.data = alloc.create_texture("gbuffer_data", SAH_FORMAT_R8G8B8A8_UNORM, W, H);
    gbuffer.emission = alloc.create_texture("gbuffer_emission", SAH_FORMAT_R8G8B8A8_SRGB, W, H);
    gbuffer.depth = alloc.create_texture("gbuffer_depth", SAH_FORMAT_D32_SFLOAT, W, H);
    TextureHandle ao = alloc.create_texture("ao", SAH_FORMAT_R32_SFLOAT, W, H);
    TextureHandle mask = alloc.create_texture("sun shadow mask", SAH_FORMAT_R32_SFLOAT, W, H);
    TextureHandle lit_scene = alloc.create_texture("lit_scene", SAH_FORMAT_R16G16B16A16_SFLOAT, W, H);
    TextureHandle antialiased = alloc.create_texture("antialiased_scene", SAH_FORMAT_R16G16B16A16_SFLOAT, W, H);
    TextureHandle swapchain = alloc.create_texture("swapchain", SAH_FORMAT_R8G8B8A8_SRGB, W, H);

    RenderScene scene;
    scene.sky.transmittance_lut = alloc.create_texture("Transmittance LUT", SAH_FORMAT_R16G16B16A16_SFLOAT, 256, 64);
    scene.sky.sky_view_lut = alloc.create_texture("Sky view LUT", SAH_FORMAT_R16G16B16A16_SFLOAT, 200, 200);
    scene.sun.set_shadow_mode((SunShadowMode)sun_mode);
    scene.sun.shadow_mask = mask;

    auto up = [&](TextureHandle t, uint32_t bpp) {
        auto blob = read_blob(in, (size_t)t->desc.width * t->desc.height * t->desc.depth * bpp);
        alloc.upload(t, blob.data(), t->desc.width * bpp);
    };
    up(gbuffer.color, 4); up(gbuffer.normals, 8); up(gbuffer.data, 4); up(gbuffer.emission, 4); up(gbuffer.depth, 4);
    up(ao, 4); up(mask, 4);
    up(scene.sky.transmittance_lut, 8); up(scene.sky.sky_view_lut, 8);

    SceneView view;  // start-up camera of the reference: scene_renderer.cpp:53-54,105-116
    view.rotate(0.f, 90.f * 3.14159265358979f / 180.f);
    view.set_position({-7.f, 1.f, 0.f});
    view.set_render_resolution(W, H);
    view.set_perspective_projection(75.f, (float)W / (float)H, 0.05f);
    view.update_transforms();

    LightPropagationVolume lpv(backend, 4, lpv_steps);
    lpv.update_cascade_transforms(view, scene.sun);
    for (int c = 0; c < 3; c++) up(lpv.get_volume(c), 8);  // stands in for RSM/VPL injection (raster, out of scope)

    LightingPhase lighting;
    lighting.set_scene(scene);
    Bloomer bloomer(backend);
    UiPhase ui;
    ui.set_resources(antialiased, swapchain);

    RenderGraph graph{backend};
    // the irradiance cache's maintenance passes through the same graph (a11): scroll cascade 0 by one cell, update three probes from
    // an all-miss trace; checked here only for "runs without error" — tests/test_probes.py holds the parity checks
    IrradianceCache cache(backend);
    cache.set_cascade(0, {-8.f, -2.f, -8.f}, 0.5f, {1.f, 0.f, 0.f});
    TextureHandle trace = alloc.create_texture("probe_trace_results", SAH_FORMAT_R16G16B16A16_SFLOAT, 20, 20, 3);
    const uint32_t probe_ids[9] = {0, 0, 0, 5, 9, 7, 31, 31, 31};
    uint32_t* probe_ids_dev = nullptr;
    if (hipMalloc(&probe_ids_dev, sizeof(probe_ids)) != hipSuccess || hipMemcpy(probe_ids_dev, probe_ids, sizeof(probe_ids), hipMemcpyHostToDevice) != hipSuccess)
        return 3;
    cache.set_trace_results(trace, probe_ids_dev, 3);
    cache.pre_render(graph, view, scene, nullptr);

    // scene_renderer.cpp:318-320, 374-401.  (lpv.pre_render() would clear the volumes for the RSM / VPL injection that follows it in the
    // reference; the uploaded volumes stand in for that injection, so only the cache's maintenance runs before the G-buffer here.)
    const uint32_t frame_count = 0;
    NoiseTexture stbn_3d_unitvec, stbn_2d_scalar;  // the blue-noise layers are not in the reference tree: empty
    IGlobalIlluminator* gi = &lpv;
    gi->post_render(graph, view, scene, gbuffer, stbn_3d_unitvec.get_layer(frame_count));

    // r.AO.Mode = Off: "clear the AO target to 1" as a compute dispatch on a scratch target (the frame's AO plane is an input of the test)
    TextureHandle ao_scratch = alloc.create_texture("ao scratch", SAH_FORMAT_R32_SFLOAT, W, H);
    const ComputePipeline clear_ao{"clear_ao", [ao_scratch](sah_ctx* ctx, const void*, const uint32_t*) {
                                       const sah_plane p = ao_scratch->plane();
                                       return sah_ao_clear(ctx, &p);
                                   }};
    ComputeDispatch<uint32_t> clear_dispatch;
    clear_dispatch.name = "Clear AO";
    clear_dispatch.num_workgroups[0] = (W + 7) / 8;
    clear_dispatch.num_workgroups[1] = (H + 7) / 8;
    clear_dispatch.compute_shader = &clear_ao;
    graph.add_compute_dispatch(clear_dispatch);
    // the same shader dispatched from an indirect buffer (render_pass.hpp:86-116) with a descriptor set, and the buffer copy verb
    TextureHandle ao_scratch2 = alloc.create_texture("ao scratch 2", SAH_FORMAT_R32_SFLOAT, W, H);
    const ComputePipeline clear_ao2{"clear_ao2", [ao_scratch2, W, H](sah_ctx* ctx, const void*, const uint32_t* groups) {
                                        if (groups[0] != (W + 7) / 8 || groups[1] != (H + 7) / 8 || groups[2] != 1) return (int)SAH_ERR_INVALID_ARGUMENT;
                                        const sah_plane p = ao_scratch2->plane();
                                        return sah_ao_clear(ctx, &p);
                                    }};
    const uint32_t counts_src[3] = {(W + 7) / 8, (H + 7) / 8, 1};
    uint32_t counts_dst[3] = {0, 0, 0};
    const Buffer counts_src_buffer{"dispatch counts (staging)", counts_src, sizeof(counts_src)}, counts_buffer{"dispatch counts", counts_dst, sizeof(counts_dst)};
    graph.add_copy_pass(BufferCopyPass{"Upload dispatch counts", &counts_buffer, &counts_src_buffer});
    IndirectComputeDispatch<uint32_t> indirect;
    indirect.name = "Clear AO (indirect)";
    indirect.descriptor_sets.push_back(DescriptorSet{{{ao_scratch2, 0x800ull /*COMPUTE_SHADER*/, 0x40ull /*SHADER_WRITE*/, 1 /*GENERAL*/}}, {}});
    indirect.dispatch = &counts_buffer;
    indirect.compute_shader = &clear_ao2;
    graph.add_compute_dispatch(indirect);
    TransitionPass to_read;
    to_read.textures.push_back({ao, kStageFragmentShader, kAccessShaderRead, kLayoutShaderReadOnly});
    graph.add_transition_pass(to_read);

    lighting.render(graph, view, gbuffer, lit_scene, ao, gi, std::nullopt, stbn_3d_unitvec, stbn_2d_scalar.get_layer(frame_count));
    gi->draw_debug_overlays(graph, view, gbuffer, lit_scene);
    evaluate_antialiasing_none(graph, lit_scene, antialiased);
    bloomer.fill_bloom_tex(graph, antialiased);
    // the same two passes as one ("Copy scene + Bloom": the copy and the first bloom dispatch in one pass over lit_scene), into a second
    // set of images that must come out bit for bit the same
    TextureHandle antialiased_fused = alloc.create_texture("antialiased_scene (fused)", SAH_FORMAT_R16G16B16A16_SFLOAT, W, H);
    Bloomer bloomer_fused(backend);
    bloomer_fused.copy_scene_and_fill_bloom_tex(graph, lit_scene, antialiased_fused);
    DynamicRenderingPass ui_pass;  // scene_renderer.cpp:426-449
    ui_pass.name = "UI";
    ui_pass.textures = {{antialiased, kStageFragmentShader, kAccessShaderRead, kLayoutShaderReadOnly},
                        {bloomer.get_bloom_tex(), kStageFragmentShader, kAccessShaderRead, kLayoutShaderReadOnly}};
    ui_pass.color_attachments = {RenderingAttachmentInfo{swapchain, 0}};
    ui_pass.execute = [&](CommandBuffer& commands) { ui.render(commands, view, bloomer.get_bloom_tex()); };
    graph.add_render_pass(std::move(ui_pass));
    // the copy verb: lit_scene into a second image, which is the one written out below
    TextureHandle lit_copy = alloc.create_texture("lit_scene copy", SAH_FORMAT_R16G16B16A16_SFLOAT, W, H);
    graph.add_copy_pass(ImageCopyPass{"Copy lit scene", lit_copy, lit_scene});
    graph.finish();
    for (const auto& e : graph.get_errors()) fprintf(stderr, "pass failed: %s\n", e.c_str());
    if (!graph.get_errors().empty()) return 1;
    {
        std::vector<float> ones((size_t)W * H);
        alloc.download(ao_scratch, ones.data(), W * 4);
        for (float v : ones)
            if (v != 1.0f) { fprintf(stderr, "Clear AO dispatch did not run\n"); return 1; }
        alloc.download(ao_scratch2, ones.data(), W * 4);
        for (float v : ones)
            if (v != 1.0f) { fprintf(stderr, "indirect Clear AO dispatch did not run\n"); return 1; }
        bool saw_set = false;
        for (const auto& u : graph.get_texture_usages()) saw_set = saw_set || u.texture == ao_scratch2;
        if (!saw_set) { fprintf(stderr, "descriptor set usages were not recorded\n"); return 1; }
    }

    {
        std::vector<unsigned char> a((size_t)W * H * 8), b((size_t)W * H * 8);
        alloc.download(antialiased, a.data(), W * 8);
        alloc.download(antialiased_fused, b.data(), W * 8);
        if (a != b) { fprintf(stderr, "Copy scene + Bloom: antialiased_scene differs from the two passes'\n"); return 1; }
        for (uint32_t m = 0; m < bloomer.get_bloom_tex()->mips.num_mips; m++) {
            const sah_plane pa = bloomer.get_bloom_tex()->mips.mips[m], pb = bloomer_fused.get_bloom_tex()->mips.mips[m];
            std::vector<unsigned char> ma((size_t)pa.width * pa.height * 8), mb(ma.size());
            if (hipMemcpy2D(ma.data(), (size_t)pa.width * 8, pa.ptr, pa.row_pitch_bytes, (size_t)pa.width * 8, pa.height, hipMemcpyDeviceToHost) != hipSuccess ||
                hipMemcpy2D(mb.data(), (size_t)pb.width * 8, pb.ptr, pb.row_pitch_bytes, (size_t)pb.width * 8, pb.height, hipMemcpyDeviceToHost) != hipSuccess) {
                fprintf(stderr, "bloom mip read-back failed\n");
                return 1;
            }
            if (ma != mb) { fprintf(stderr, "Copy scene + Bloom: bloom mip %u differs from the two passes'\n", m); return 1; }
        }
    }

    FILE* out = fopen(argv[2], "wb");
    if (!out) { perror("open output"); return 2; }
    fwrite(&view.get_gpu_data(), sizeof(sah_view_data), 1, out);
    fwrite(&scene.sun.get_constants(), sizeof(sah_sun_light_constants), 1, out);
    fwrite(lpv.get_cascade_matrices(), sizeof(sah_lpv_cascade_matrices), 4, out);
    std::vector<unsigned char> buf((size_t)W * H * 8);
    for (int c = 0; c < 3; c++) {  // propagated LPV (A volumes)
        std::vector<unsigned char> v((size_t)128 * 32 * 32 * 8);
        alloc.download(lpv.get_volume(c), v.data(), 128 * 8);
        fwrite(v.data(), 1, v.size(), out);
    }
    alloc.download(lit_copy, buf.data(), W * 8);
    fwrite(buf.data(), 1, (size_t)W * H * 8, out);
    alloc.download(swapchain, buf.data(), W * 4);
    fwrite(buf.data(), 1, (size_t)W * H * 4, out);
    fclose(out);
    return 0;
}
