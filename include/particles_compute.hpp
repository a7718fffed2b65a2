// particles_compute.hpp — C++ host-side mirror of the reference's `Compute`
// (src/compute.rs:11-228) over the C ABI of particles_hip.h.
//
// The reference's host is Rust; where no Rust toolchain is available this
// header is the compiled-language host layer: same method names, argument
// meaning and call order as compute.rs, with the wgpu `Device` /
// `ComputePass` arguments replaced by the HIP device ordinal given at
// construction.  The reference has no error returns (wgpu validation failures
// panic); here every failure throws particles::Error carrying the status code
// and pcl_last_error().
#pragma once

#include "particles_hip.h"

#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace particles {

using Particle = pcl_particle;      // src/main.rs:76-83
using SimParams = pcl_sim_params;   // src/sim_params.rs:6-17
using Poly7 = pcl_poly7;            // src/poly7.rs:6-10
using Diagnostics = pcl_diagnostics;
using Cell = float[4];              // [x, y, z, 1.0]: src/grid.rs:97-103

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

inline void check(int rc)
{
    if (rc != PCL_OK) throw Error(rc, std::string("particles_hip: ") + pcl_last_error());
}

// SimParams::new(): src/sim_params.rs:20-31.
inline SimParams default_sim_params()
{
    SimParams p;
    pcl_sim_params_default(&p);
    return p;
}

class Compute {
public:
    // Compute::new(device, particles, force_grid): src/compute.rs:25.  Copies
    // both inputs; parameters start as SimParams::new() (delta_t = 0).
    Compute(const Particle* particles, std::size_t n, const Cell* force_grid, std::size_t n_cells, int device = 0)
    {
        check(pcl_create(particles, n, force_grid, n_cells, device, &h_));
    }
    Compute(const std::vector<Particle>& particles, const std::vector<float>& force_grid_xyzw, int device = 0)
        : Compute(particles.data(), particles.size(), reinterpret_cast<const Cell*>(force_grid_xyzw.data()),
                  force_grid_xyzw.size() / 4, device)
    {
    }
    ~Compute() { pcl_destroy(h_); }
    Compute(const Compute&) = delete;
    Compute& operator=(const Compute&) = delete;
    Compute(Compute&& o) noexcept : h_(std::exchange(o.h_, nullptr)) {}
    Compute& operator=(Compute&& o) noexcept
    {
        if (this != &o) {
            pcl_destroy(h_);
            h_ = std::exchange(o.h_, nullptr);
        }
        return *this;
    }

    // pub num_particles: src/compute.rs:19.
    std::size_t num_particles() const { return pcl_num_particles(h_); }

    // upload_particles: src/compute.rs:182-196.
    void upload_particles(const Particle* particles, std::size_t n) { check(pcl_upload_particles(h_, particles, n)); }
    void upload_particles(const std::vector<Particle>& p) { upload_particles(p.data(), p.size()); }

    // update_force_grid: src/compute.rs:198-204.
    void update_force_grid(const Cell* force_grid, std::size_t n_cells)
    {
        check(pcl_update_force_grid(h_, force_grid, n_cells));
    }
    // Brush edit of cells [first_cell, first_cell + count) instead of the
    // whole-field re-upload of src/main.rs:371-380.
    void patch_force_grid(std::size_t first_cell, const Cell* cells, std::size_t count)
    {
        check(pcl_patch_force_grid(h_, first_cell, cells, count));
    }

    // update_sim_params: src/compute.rs:206-219 (delta_t travels in the block).
    void update_sim_params(const SimParams& p) { check(pcl_update_sim_params(h_, &p)); }

    // compute(): src/compute.rs:221-227 — one update, asynchronous.
    void compute() { check(pcl_step(h_)); }
    // n updates with the result of n compute() calls; consecutive streaming steps share launches.
    void compute_n(std::uint64_t n_steps) { check(pcl_step_n(h_, n_steps)); }
    // step(dt): set delta_t and run one update.
    void step(float dt) { check(pcl_step_dt(h_, dt)); }

    // Read back the latest state in caller order (the reference only exposes
    // the buffers to the renderer: src/compute.rs:13, src/renderer.rs:253-257).
    std::vector<Particle> read_particles(std::size_t first, std::size_t count)
    {
        std::vector<Particle> out(count);
        check(pcl_read_particles(h_, out.data(), first, count));
        return out;
    }
    std::vector<Particle> read_particles() { return read_particles(0, num_particles()); }

    // Renderer interop: 48-byte instance records into a device buffer.
    void pack_particles_device(void* dev_out, std::size_t first, std::size_t count)
    {
        check(pcl_pack_particles_device(h_, dev_out, first, count));
    }

    // The pair loop over ranges held by several handles: see particles_hip.h.
    void pack_pair_sources_device(void* dev_out, std::size_t capacity_records)
    {
        check(pcl_pack_pair_sources_device(h_, dev_out, capacity_records));
    }
    void compute_gathered(const void* dev_sources, std::size_t n_sources)
    {
        check(pcl_step_pairs_gathered(h_, dev_sources, n_sources));
    }

    // All records in memory order (an instance stream does not care): one streaming pass.
    void pack_particles_device_unordered(void* dev_out) { check(pcl_pack_particles_device_unordered(h_, dev_out)); }

    Diagnostics diagnostics()
    {
        Diagnostics d;
        check(pcl_diagnostics_read(h_, &d));
        return d;
    }
    void enable_step_diagnostics(bool on) { check(pcl_enable_step_diagnostics(h_, on ? 1 : 0)); }
    void set_reorder_interval(int interval) { check(pcl_set_reorder_interval(h_, interval)); }
    void synchronize() { check(pcl_synchronize(h_)); }
    void set_stream(void* hip_stream) { check(pcl_set_stream(h_, hip_stream)); }
    void use_null_stream() { check(pcl_use_null_stream(h_)); }
    int last_step_kernel() const { return pcl_last_step_kernel(h_); }
    pcl_compute* handle() { return h_; }

private:
    pcl_compute* h_ = nullptr;
};

}  // namespace particles
