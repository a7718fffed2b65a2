// Walks the whole C ABI of include/particles_hip.h once on GPU 0, with sets large
// enough that the tile order, its refreshes, both pair sweeps, the brush and the
// one-rank communicator all run.  It checks status codes, not results (the Python
// tests compare results with the oracle): its purpose is to run the library's HOST
// code under AddressSanitizer and UndefinedBehaviorSanitizer (`make sanitize`
// builds host-instrumented copies of the library and of this program; device code
// is not instrumented).  Exit status 0 = every call returned what was expected.
#include "particles_hip.h"

#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

int g_failures = 0;

void expect(int got, int want, const char* what, int line)
{
    if (got != want) {
        std::fprintf(stderr, "api_walk.cpp:%d: %s returned %d, expected %d (%s)\n", line, what, got, want, pcl_last_error());
        ++g_failures;
    }
}

#define OK(call) expect((call), PCL_OK, #call, __LINE__)
#define FAILS(call, code) expect((call), (code), #call, __LINE__)
#define HIP_OK(call)                                                                     \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) {                                                          \
            std::fprintf(stderr, "api_walk.cpp:%d: %s: %s\n", __LINE__, #call, hipGetErrorString(e_)); \
            std::exit(3);                                                                \
        }                                                                                \
    } while (0)

uint32_t lcg(uint32_t& s)
{
    s = s * 1664525u + 1013904223u;
    return s >> 8;
}

float unit(uint32_t& s) { return (float)lcg(s) / 16777216.0f; }

std::vector<pcl_particle> random_particles(size_t n, float extent, uint32_t seed)
{
    std::vector<pcl_particle> p(n);
    uint32_t s = seed;
    for (size_t i = 0; i < n; ++i) {
        std::memset(&p[i], 0, sizeof(pcl_particle));
        for (int a = 0; a < 3; ++a) {
            p[i].pos[a] = extent * (unit(s) - 0.5f);
            p[i].vel[a] = 10.0f * (unit(s) - 0.5f);
        }
        p[i].pos[3] = p[i].vel[3] = 1.0f;
        p[i].ty = (uint32_t)(i % 5);
    }
    return p;
}

std::vector<float> swirl(size_t nx, size_t ny)
{
    std::vector<float> g(nx * ny * 4);
    for (size_t ix = 0; ix < nx; ++ix)
        for (size_t iy = 0; iy < ny; ++iy) {
            float* c = &g[4 * (ix * ny + iy)];
            const float x = ((float)ix + 0.5f) / (float)nx - 0.5f, y = ((float)iy + 0.5f) / (float)ny - 0.5f;
            c[0] = -y * 2.0f;
            c[1] = x * 2.0f;
            c[2] = 0.0f;
            c[3] = 1.0f;
        }
    return g;
}

typedef const float (*cells_t)[4];
cells_t as_cells(const std::vector<float>& v) { return reinterpret_cast<cells_t>(v.data()); }

void walk_streaming(pcl_compute* h, const std::vector<float>& field, size_t n_cells, const pcl_sim_params& sp)
{
    const size_t n = ((size_t)1 << 20) + 3;
    OK(pcl_generate_particles(h, n, 12345, 0, 20.0f, 10.0f));
    OK(pcl_update_sim_params(h, &sp));
    OK(pcl_step_timing_begin(h, 64));
    OK(pcl_timer_start(h));
    for (int s = 0; s < 30; ++s) OK(pcl_step(h));
    OK(pcl_enable_step_diagnostics(h, 1));
    for (int s = 0; s < 30; ++s) OK(pcl_step(h));
    float ms = 0.0f;
    OK(pcl_timer_stop(h, &ms));
    std::vector<float> launch_ms(64);
    std::vector<int32_t> kinds(64);
    size_t recorded = 0;
    OK(pcl_step_timing_read(h, launch_ms.data(), kinds.data(), 64, &recorded));
    if (recorded != 60) { std::fprintf(stderr, "recorded %zu launches, expected 60\n", recorded); ++g_failures; }
    if (pcl_reorder_count(h) < 2) { std::fprintf(stderr, "no refresh of the tile order ran\n"); ++g_failures; }
    pcl_diagnostics d;
    OK(pcl_diagnostics_read(h, &d));
    if (d.count != (double)n) { std::fprintf(stderr, "diagnostics count %.0f\n", d.count); ++g_failures; }
    OK(pcl_step_n(h, 40));
    OK(pcl_enable_step_diagnostics(h, 0));
    OK(pcl_step_dt(h, 1.0f / 120.0f));
    OK(pcl_update_sim_params(h, &sp));

    // reads: whole, ragged ranges, empty range, arrays left out
    std::vector<pcl_particle> out(n);
    OK(pcl_read_particles(h, out.data(), 0, n));
    OK(pcl_read_particles(h, out.data(), 12345, 777));
    OK(pcl_read_particles(h, out.data(), n, 0));
    FAILS(pcl_read_particles(h, out.data(), n - 1, 2), PCL_ERR_OUT_OF_RANGE);
    std::vector<float> x(n), vz(n);
    std::vector<uint8_t> ty(n);
    OK(pcl_read_soa(h, 0, n, x.data(), nullptr, nullptr, nullptr, nullptr, vz.data(), ty.data()));
    OK(pcl_read_soa(h, 5, 1001, x.data(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr));
    std::vector<uint32_t> idx(n);
    OK(pcl_debug_cell_indices(h, idx.data(), 0, n));
    OK(pcl_debug_cell_indices(h, idx.data(), 3, 100));
    OK(pcl_debug_slot_order(h, idx.data(), n));
    FAILS(pcl_debug_slot_order(h, idx.data(), n - 1), PCL_ERR_INVALID);
    uint64_t sum = 0, x_or = 0;
    OK(pcl_debug_state_checksum(h, 7, &sum, &x_or));

    // device-side consumers
    void* dev = nullptr;
    HIP_OK(hipMalloc(&dev, n * sizeof(pcl_particle)));
    OK(pcl_pack_particles_device(h, dev, 0, n));
    OK(pcl_pack_particles_device(h, dev, 1000, 4097));
    OK(pcl_pack_particles_device_unordered(h, dev));
    OK(pcl_diagnostics_device(h, dev));
    OK(pcl_diagnostics_device_noalloc(h, dev));
    OK(pcl_synchronize(h));

    // settings and reports
    uint32_t grid = 0, block = 0, per_lane = 0, intended = 0, granted = 0;
    OK(pcl_launch_geometry(h, &grid, &block, &per_lane));
    OK(pcl_launch_residency(h, &intended, &granted));
    char pci[32];
    OK(pcl_device_pci_bus_id(h, pci, sizeof pci));
    FAILS(pcl_device_pci_bus_id(h, pci, 4), PCL_ERR_INVALID);
    OK(pcl_reorder_now(h));
    OK(pcl_set_reorder_interval(h, 0));
    for (int s = 0; s < 3; ++s) OK(pcl_step(h));
    OK(pcl_set_reorder_interval(h, 5));
    for (int s = 0; s < 12; ++s) OK(pcl_step(h));
    OK(pcl_set_reorder_interval(h, -1));
    (void)pcl_reorder_interval_in_use(h);
    for (uint32_t variant : {0x4u, 0x100u, 0x8u, 0x1u, 0x400000u, 0x13000000u, 0u}) {
        OK(pcl_set_tuning(h, 0, variant));
        for (int s = 0; s < 26; ++s) OK(pcl_step(h));
    }
    OK(pcl_set_tuning(h, 64, 0));
    for (int s = 0; s < 3; ++s) OK(pcl_step(h));
    OK(pcl_set_tuning(h, 0, 0));

    // field edits: whole, range, list with a repeated index, from device memory, brush
    OK(pcl_update_force_grid(h, as_cells(field), n_cells));
    OK(pcl_patch_force_grid(h, 100, as_cells(field), 333));
    FAILS(pcl_patch_force_grid(h, n_cells - 1, as_cells(field), 2), PCL_ERR_OUT_OF_RANGE);
    std::vector<uint32_t> cells_idx = {5, 9, 5, (uint32_t)n_cells - 1, 9, 5, 0};
    OK(pcl_patch_force_grid_indexed(h, cells_idx.data(), as_cells(field), cells_idx.size()));
    OK(pcl_patch_force_grid_indexed(h, cells_idx.data(), as_cells(field), 0));
    cells_idx[2] = (uint32_t)n_cells;
    FAILS(pcl_patch_force_grid_indexed(h, cells_idx.data(), as_cells(field), cells_idx.size()), PCL_ERR_OUT_OF_RANGE);
    cells_idx[2] = 5;
    void* dev_cells = nullptr;
    void* dev_idx = nullptr;
    HIP_OK(hipMalloc(&dev_cells, n_cells * 16));
    HIP_OK(hipMalloc(&dev_idx, cells_idx.size() * 4));
    HIP_OK(hipMemcpy(dev_cells, field.data(), n_cells * 16, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dev_idx, cells_idx.data(), cells_idx.size() * 4, hipMemcpyHostToDevice));
    OK(pcl_update_force_grid_device(h, dev_cells, n_cells));
    OK(pcl_patch_force_grid_device(h, 17, dev_cells, 1000));
    OK(pcl_patch_force_grid_indexed_device(h, dev_idx, dev_cells, cells_idx.size()));
    pcl_brush_input in;
    pcl_brush_input_default(&in);
    in.falloff_dist = 1.5f;
    for (int mode = 0; mode < 3; ++mode)
        for (int falloff = 0; falloff < 3; ++falloff) {
            in.mode = mode;
            in.falloff = falloff;
            in.relative = (mode + falloff) & 1;
            in.cursor_pos[0] = 0.3f * (float)mode;
            OK(pcl_brush_mouse_down(h, &in));
            in.cursor_pos[1] = 0.2f;
            OK(pcl_brush_mouse_moved(h, &in));
            size_t selected = 0;
            std::vector<uint32_t> sel(16);
            OK(pcl_brush_selection(h, sel.data(), sel.size(), &selected));
            OK(pcl_brush_selection(h, nullptr, 0, &selected));
            OK(pcl_brush_mouse_up(h));
            in.cursor_pos[1] = 0.0f;
        }
    in.mode = 99;
    FAILS(pcl_brush_mouse_down(h, &in), PCL_ERR_INVALID);
    in.mode = PCL_EDIT_SHIFT;
    void* dev_in = nullptr;
    HIP_OK(hipMalloc(&dev_in, sizeof in));
    HIP_OK(hipMemcpy(dev_in, &in, sizeof in, hipMemcpyHostToDevice));
    OK(pcl_brush_mouse_down_device(h, dev_in));
    OK(pcl_brush_mouse_moved_device(h, dev_in));
    OK(pcl_brush_mouse_up(h));
    for (int s = 0; s < 3; ++s) OK(pcl_step(h));
    std::vector<float> back(n_cells * 4);
    OK(pcl_read_force_grid(h, reinterpret_cast<float(*)[4]>(back.data()), 0, n_cells));
    OK(pcl_read_force_grid(h, reinterpret_cast<float(*)[4]>(back.data()), n_cells - 3, 3));
    FAILS(pcl_read_force_grid(h, reinterpret_cast<float(*)[4]>(back.data()), n_cells, 1), PCL_ERR_OUT_OF_RANGE);
    if (pcl_num_cells(h) != n_cells) { std::fprintf(stderr, "pcl_num_cells\n"); ++g_failures; }

    // a caller's stream, the null stream, back to the internal one
    hipStream_t stream = nullptr;
    HIP_OK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    OK(pcl_synchronize(h));
    OK(pcl_set_stream(h, stream));
    for (int s = 0; s < 3; ++s) OK(pcl_step(h));
    OK(pcl_synchronize(h));
    OK(pcl_use_null_stream(h));
    for (int s = 0; s < 3; ++s) OK(pcl_step(h));
    OK(pcl_synchronize(h));
    OK(pcl_set_stream(h, nullptr));
    OK(pcl_step(h));
    OK(pcl_synchronize(h));
    HIP_OK(hipStreamDestroy(stream));
    HIP_OK(hipFree(dev));
    HIP_OK(hipFree(dev_cells));
    HIP_OK(hipFree(dev_idx));
    HIP_OK(hipFree(dev_in));
}

void walk_pairs(pcl_compute* h, pcl_sim_params sp)
{
    uint32_t s = 99;
    for (int k = 0; k < PCL_NUM_POLYS; ++k)
        for (int c = 0; c < 8; ++c) sp.attraction_force[k].coeffs[c] = 2.0f * unit(s) - 1.0f;
    sp.cut_off_distance = 1.0f;
    OK(pcl_update_sim_params(h, &sp));

    // exact sweep, ragged size
    std::vector<pcl_particle> p = random_particles(5003, 4.0f, 7);
    OK(pcl_upload_particles(h, p.data(), p.size()));
    OK(pcl_set_pair_mode(h, PCL_PAIRS_ALL));
    for (int i = 0; i < 3; ++i) OK(pcl_step(h));
    if (pcl_last_step_kernel(h) != PCL_KERNEL_PAIRS) { std::fprintf(stderr, "pair kernel did not run\n"); ++g_failures; }
    sp.distance_exponent = -1.5f;
    OK(pcl_update_sim_params(h, &sp));
    OK(pcl_step(h));
    sp.distance_exponent = 0.0f;
    OK(pcl_update_sim_params(h, &sp));

    // gathered partners: this handle's own records padded to a capacity
    const size_t cap = 6000;
    void* dev = nullptr;
    HIP_OK(hipMalloc(&dev, cap * 16));
    OK(pcl_pack_pair_sources_device(h, dev, cap));
    OK(pcl_step_pairs_gathered(h, dev, cap));
    FAILS(pcl_pack_pair_sources_device(h, dev, 10), PCL_ERR_INVALID);
    OK(pcl_synchronize(h));
    HIP_OK(hipFree(dev));

    // cell-binned sweeps: spread out (per-lane sweep) and clustered (shared sweep), both forced too
    for (float extent : {20.0f, 1.0f}) {
        p = random_particles(40001, extent, 11);
        OK(pcl_upload_particles(h, p.data(), p.size()));
        OK(pcl_set_pair_mode(h, PCL_PAIRS_BINNED));
        for (uint32_t variant : {0u, 0x800000u, 0x80000000u, 0x10000u, 0x20000u}) {
            OK(pcl_set_tuning(h, 0, variant));
            for (int i = 0; i < 4; ++i) OK(pcl_step(h));
        }
        OK(pcl_set_tuning(h, 0, 0));
        uint32_t shared = 0, occupied = 0;
        OK(pcl_pair_sweep_info(h, &shared, &occupied));
        std::vector<uint32_t> cell(p.size()), ends(64 * 64 * 64);
        uint32_t per_axis = 0, reach = 0;
        OK(pcl_debug_pair_cells(h, cell.data(), ends.data(), ends.size(), &per_axis, &reach));
        OK(pcl_debug_pair_cells(h, cell.data(), nullptr, 0, &per_axis, &reach));
        std::vector<pcl_particle> out(p.size());
        OK(pcl_read_particles(h, out.data(), 0, out.size()));
    }
    OK(pcl_set_pair_mode(h, PCL_PAIRS_AUTO));
    FAILS(pcl_set_pair_mode(h, 9), PCL_ERR_INVALID);
    OK(pcl_force_pair_kernel(h, 1));
    OK(pcl_step(h));
    OK(pcl_force_pair_kernel(h, 0));

    // SoA upload, a refused upload, an empty set
    const size_t n = 1000;
    std::vector<float> a(n, 0.25f);
    std::vector<uint8_t> ty(n, 3);
    OK(pcl_upload_particles_soa(h, n, a.data(), a.data(), a.data(), a.data(), a.data(), a.data(), ty.data()));
    ty[17] = 5;
    FAILS(pcl_upload_particles_soa(h, n, a.data(), a.data(), a.data(), a.data(), a.data(), a.data(), ty.data()), PCL_ERR_INVALID);
    p = random_particles(10, 1.0f, 3);
    p[4].ty = 77;
    FAILS(pcl_upload_particles(h, p.data(), p.size()), PCL_ERR_INVALID);
    OK(pcl_upload_particles(h, nullptr, 0));
    OK(pcl_step(h));
    pcl_diagnostics d;
    OK(pcl_diagnostics_read(h, &d));
    if (pcl_num_particles(h) != 0 || d.count != 0.0) { std::fprintf(stderr, "empty set\n"); ++g_failures; }
    (void)pcl_step_count(h);
}

void walk_comm(pcl_compute* h, const std::vector<float>& field, size_t n_cells, pcl_sim_params sp)
{
    unsigned char id[PCL_UNIQUE_ID_BYTES];
    FAILS(pcl_comm_step(h), PCL_ERR_INVALID);
    const int rc = pcl_comm_unique_id(id);
    if (rc == PCL_ERR_COMM) {
        std::fprintf(stderr, "api_walk: RCCL not loadable here (%s): communicator calls skipped\n", pcl_last_error());
        return;
    }
    OK(rc);
    OK(pcl_comm_attach(h, id, 0, 1));
    if (pcl_comm_rank(h) != 0 || pcl_comm_world(h) != 1 || pcl_comm_count(h) != 1) { std::fprintf(stderr, "communicator geometry\n"); ++g_failures; }
    OK(pcl_generate_particles(h, (size_t)1 << 20, 5, 0, 20.0f, 10.0f));
    OK(pcl_update_sim_params(h, &sp));
    OK(pcl_comm_update_force_grid(h, as_cells(field), n_cells, 0));
    OK(pcl_comm_patch_force_grid(h, 10, as_cells(field), 100, 0));
    const uint32_t idx[4] = {1, 2, 1, 3};
    OK(pcl_comm_patch_force_grid_indexed(h, idx, as_cells(field), 4, 0));
    pcl_brush_input in;
    pcl_brush_input_default(&in);
    OK(pcl_comm_brush_mouse_down(h, &in, 0));
    for (int e = 0; e < 20; ++e) {           // more events than the ring of pinned records holds
        in.cursor_pos[0] = 0.1f * (float)e;
        OK(pcl_comm_brush_mouse_moved(h, &in, 0));
        OK(pcl_comm_step(h));
    }
    OK(pcl_brush_mouse_up(h));
    for (size_t count : {(size_t)3, (size_t)700, (size_t)1, (size_t)2000, (size_t)40}) {   // pinned staging grows and is reused
        std::vector<uint32_t> list(count);
        for (size_t k = 0; k < count; ++k) list[k] = (uint32_t)((k * 7919u) % n_cells);
        OK(pcl_comm_patch_force_grid_indexed(h, list.data(), as_cells(field), count, 0));
        OK(pcl_comm_step(h));
    }
    OK(pcl_enable_step_diagnostics(h, 1));
    for (int s = 0; s < 30; ++s) OK(pcl_comm_step(h));
    pcl_diagnostics d;
    OK(pcl_comm_diagnostics_read(h, &d));
    const void* dev_record = nullptr;
    OK(pcl_comm_diagnostics_device(h, &dev_record));
    OK(pcl_enable_step_diagnostics(h, 0));
    // pair term active: the all-gather path
    for (int c = 0; c < 8; ++c) sp.attraction_force[3].coeffs[c] = 0.5f;
    OK(pcl_update_sim_params(h, &sp));
    std::vector<pcl_particle> p = random_particles(3001, 4.0f, 21);
    OK(pcl_upload_particles(h, p.data(), p.size()));
    OK(pcl_set_pair_mode(h, PCL_PAIRS_ALL));
    for (int s = 0; s < 3; ++s) OK(pcl_comm_step(h));
    OK(pcl_synchronize(h));
    OK(pcl_comm_detach(h));
    if (pcl_comm_world(h) != 0) { std::fprintf(stderr, "detach\n"); ++g_failures; }
    OK(pcl_set_pair_mode(h, PCL_PAIRS_AUTO));
}

void walk_group(const std::vector<float>& field, size_t n_cells, const pcl_sim_params& sp)
{
    const int devices[1] = {0};
    std::vector<pcl_particle> p = random_particles(70001, 20.0f, 31);
    pcl_group* g = nullptr;
    const int rc = pcl_group_create(devices, 1, p.data(), p.size(), as_cells(field), n_cells, &g);
    if (rc == PCL_ERR_COMM) {
        std::fprintf(stderr, "api_walk: RCCL not loadable here: group calls skipped\n");
        return;
    }
    OK(rc);
    if (pcl_group_size(g) != 1 || pcl_group_num_particles(g) != p.size() || pcl_group_handle(g, 0) == nullptr) { std::fprintf(stderr, "group geometry\n"); ++g_failures; }
    OK(pcl_group_update_sim_params(g, &sp));
    OK(pcl_group_enable_step_diagnostics(g, 1));
    for (int s = 0; s < 5; ++s) OK(pcl_group_step(g));
    OK(pcl_group_update_force_grid(g, as_cells(field), n_cells));
    OK(pcl_group_patch_force_grid(g, 3, as_cells(field), 50));
    const uint32_t idx[3] = {4, 4, 8};
    OK(pcl_group_patch_force_grid_indexed(g, idx, as_cells(field), 3));
    pcl_brush_input in;
    pcl_brush_input_default(&in);
    OK(pcl_group_brush_mouse_down(g, &in));
    OK(pcl_group_brush_mouse_moved(g, &in));
    OK(pcl_group_brush_mouse_up(g));
    pcl_diagnostics d;
    OK(pcl_group_diagnostics_read(g, &d));
    std::vector<pcl_particle> out(p.size());
    OK(pcl_group_read_particles(g, out.data(), 0, out.size()));
    OK(pcl_group_read_particles(g, out.data(), 69990, 11));
    std::vector<float> back(n_cells * 4);
    OK(pcl_group_read_force_grid(g, 0, reinterpret_cast<float(*)[4]>(back.data()), 0, n_cells));
    OK(pcl_group_upload_particles(g, p.data(), 1234));
    OK(pcl_group_generate_particles(g, (size_t)1 << 18, 9, 20.0f, 10.0f));
    for (int s = 0; s < 3; ++s) OK(pcl_group_step(g));
    OK(pcl_group_synchronize(g));
    OK(pcl_group_destroy(g));
}

void walk_authoring()
{
    float pts[8][2];
    for (int i = 0; i < 8; ++i) {
        pts[i][0] = 0.25f * (float)i;
        pts[i][1] = std::sin((float)i);
    }
    pcl_poly7 poly;
    OK(pcl_poly7_from_points(pts, &poly));
    (void)pcl_poly7_eval(&poly, 0.7f);
    pts[3][0] = pts[2][0];
    FAILS(pcl_poly7_from_points(pts, &poly), PCL_ERR_SINGULAR);
    std::vector<float> cells(6 * 5 * 4 * 4);
    pcl_grid_centered(6, 5, 4, reinterpret_cast<float(*)[4]>(cells.data()));
    const float v[3] = {1.0f, 2.0f, 3.0f}, pos[3] = {-10.0f, -10.0f, -10.0f}, dir[3] = {20.0f, 20.0f, 20.0f};
    pcl_grid_uniform(6, 5, 4, v, reinterpret_cast<float(*)[4]>(cells.data()));
    pcl_grid_positions(6, 5, 4, pos, dir, reinterpret_cast<float(*)[4]>(cells.data()));
    pcl_sim_params sp;
    pcl_sim_params_default(&sp);
    std::vector<float> field(1000 * 4);
    OK(pcl_sim_params_force_grid(&sp, 1, reinterpret_cast<float(*)[4]>(field.data()), 1000));
    OK(pcl_sim_params_force_grid(&sp, 0, reinterpret_cast<float(*)[4]>(field.data()), 1000));
    FAILS(pcl_sim_params_force_grid(&sp, 1, reinterpret_cast<float(*)[4]>(field.data()), 999), PCL_ERR_OUT_OF_RANGE);
}

// "Different handles are independent": four threads, each with a handle of its own on the same GPU, make the
// same calls at the same time; the same seed must give every thread the same state.
void walk_threads(const std::vector<float>& field, size_t n_cells, const pcl_sim_params& sp)
{
    const int n_threads = 4;
    std::vector<uint64_t> sums(n_threads, 0), xors(n_threads, 0);
    std::vector<int> failed(n_threads, 0);
    std::vector<std::thread> threads;
    for (int t = 0; t < n_threads; ++t)
        threads.emplace_back([&, t] {
            pcl_compute* h = nullptr;
            int bad = 0;
            bad += pcl_create(nullptr, 0, as_cells(field), n_cells, 0, &h) != PCL_OK;
            if (!h) { failed[t] = 1; return; }
            bad += pcl_generate_particles(h, ((size_t)1 << 20) + 1, 77, 0, 20.0f, 10.0f) != PCL_OK;
            bad += pcl_update_sim_params(h, &sp) != PCL_OK;
            bad += pcl_enable_step_diagnostics(h, t & 1) != PCL_OK;
            for (int s = 0; s < 50; ++s) bad += pcl_step(h) != PCL_OK;
            bad += pcl_step_n(h, 20) != PCL_OK;
            pcl_diagnostics d;
            bad += pcl_diagnostics_read(h, &d) != PCL_OK;
            bad += pcl_debug_state_checksum(h, 0, &sums[t], &xors[t]) != PCL_OK;
            bad += pcl_step(nullptr) != PCL_ERR_INVALID;            // the error text is per thread
            bad += std::strlen(pcl_last_error()) == 0;
            bad += pcl_destroy(h) != PCL_OK;
            failed[t] = bad;
        });
    for (std::thread& th : threads) th.join();
    for (int t = 0; t < n_threads; ++t) {
        if (failed[t]) { std::fprintf(stderr, "thread %d: %d calls failed\n", t, failed[t]); ++g_failures; }
        if (sums[t] != sums[0] || xors[t] != xors[0]) { std::fprintf(stderr, "thread %d ended in another state than thread 0\n", t); ++g_failures; }
    }
}

}  // namespace

int main()
{
    if (pcl_device_count() < 1) {
        std::fprintf(stderr, "api_walk: no GPU\n");
        return 2;
    }
    walk_authoring();

    const size_t nx = 128, ny = 128, n_cells = nx * ny;
    const std::vector<float> field = swirl(nx, ny);
    pcl_sim_params sp;
    pcl_sim_params_default(&sp);
    sp.force_grid_dimensions[0] = (uint32_t)nx;
    sp.force_grid_dimensions[1] = (uint32_t)ny;
    sp.force_grid_dimensions[2] = 1;
    sp.delta_t = 1.0f / 60.0f;
    const float masses[5] = {1.0f, 0.5f, 2.0f, 1.5f, 0.75f};
    for (int t = 0; t < 5; ++t) sp.particle_type_masses[t].mass = masses[t];

    // null handles and null outputs come back as status codes
    FAILS(pcl_step(nullptr), PCL_ERR_INVALID);
    FAILS(pcl_create(nullptr, 0, as_cells(field), n_cells, 0, nullptr), PCL_ERR_INVALID);
    FAILS(pcl_create(nullptr, 3, as_cells(field), n_cells, 0, nullptr), PCL_ERR_INVALID);
    OK(pcl_destroy(nullptr));

    pcl_compute* h = nullptr;
    OK(pcl_create(nullptr, 0, as_cells(field), n_cells, 0, &h));
    if (!h) return 1;
    OK(pcl_step(h));                       // no particles: nothing to do
    walk_streaming(h, field, n_cells, sp);
    walk_pairs(h, sp);
    walk_comm(h, field, n_cells, sp);
    OK(pcl_destroy(h));
    walk_group(field, n_cells, sp);

    walk_threads(field, n_cells, sp);

    // two handles alive at once, destroyed in the other order
    pcl_compute *a = nullptr, *b = nullptr;
    std::vector<pcl_particle> p = random_particles(1000, 4.0f, 5);
    OK(pcl_create(p.data(), p.size(), as_cells(field), n_cells, 0, &a));
    OK(pcl_create(p.data(), p.size(), as_cells(field), n_cells, 0, &b));
    OK(pcl_update_sim_params(a, &sp));
    OK(pcl_update_sim_params(b, &sp));
    OK(pcl_step(a));
    OK(pcl_step(b));
    OK(pcl_destroy(a));
    OK(pcl_step(b));
    OK(pcl_destroy(b));

    if (g_failures) {
        std::fprintf(stderr, "api_walk: %d unexpected status codes\n", g_failures);
        return 1;
    }
    std::printf("api_walk ok\n");
    std::fflush(stdout);   // on record even if a library's exit handler ends the process
    return 0;
}
