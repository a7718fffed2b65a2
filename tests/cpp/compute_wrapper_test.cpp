// Drives the C++ host mirror (include/particles_compute.hpp) the way the
// reference's App drives Compute (src/main.rs:223-255, 333-390): construct from
// the start-up lattice and the centred field, then per frame update_force_grid,
// update_sim_params, compute.  Writes the final particle records to the file
// named on the command line; the Python test compares them with the oracle.
#include "particles_compute.hpp"

#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char** argv)
{
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s <out.bin> <steps>\n", argv[0]);
        return 2;
    }
    const int steps = std::atoi(argv[2]);
    try {
        particles::SimParams sp = particles::default_sim_params();
        const std::size_t nx = sp.force_grid_dimensions[0], ny = sp.force_grid_dimensions[1], nz = sp.force_grid_dimensions[2];
        // ParticleSystem::new((5,2,2), 10, 10, 10): src/main.rs:117-154
        std::vector<particles::Particle> p(nx * ny * nz);
        for (std::size_t ix = 0; ix < nx; ++ix)
            for (std::size_t iy = 0; iy < ny; ++iy)
                for (std::size_t iz = 0; iz < nz; ++iz) {
                    const std::size_t i = ix * ny * nz + iy * nz + iz;
                    particles::Particle q = {};
                    q.pos[0] = ((float)ix / (float)nx) * 5.0f;
                    q.pos[1] = ((float)iy / (float)ny) * 2.0f;
                    q.pos[2] = ((float)iz / (float)nz) * 2.0f;
                    q.pos[3] = 1.0f;
                    q.ty = (uint32_t)(i % 5);
                    p[i] = q;
                }
        // Grid::new_centered + get_force_vectors: src/grid.rs:68-103
        std::vector<float> g(nx * ny * nz * 4);
        std::size_t k = 0;
        for (std::size_t ix = 0; ix < nx; ++ix)
            for (std::size_t iy = 0; iy < ny; ++iy)
                for (std::size_t iz = 0; iz < nz; ++iz, ++k) {
                    g[4 * k + 0] = (float)nx * 0.5f - (float)ix;
                    g[4 * k + 1] = (float)ny * 0.5f - (float)iy;
                    g[4 * k + 2] = (float)nz * 0.5f - (float)iz;
                    g[4 * k + 3] = 1.0f;
                }
        particles::Compute compute(p, g, 0);
        sp.delta_t = 1.0f / 60.0f;
        for (int s = 0; s < steps; ++s) {
            compute.update_force_grid(reinterpret_cast<const particles::Cell*>(g.data()), g.size() / 4);
            compute.update_sim_params(sp);
            compute.compute();
        }
        const std::vector<particles::Particle> out = compute.read_particles();
        const particles::Diagnostics d = compute.diagnostics();
        std::FILE* f = std::fopen(argv[1], "wb");
        if (!f) return 3;
        std::fwrite(out.data(), sizeof(particles::Particle), out.size(), f);
        std::fclose(f);
        std::printf("%zu particles, kinetic energy %.9g\n", out.size(), d.kinetic_energy);
        // error path: a type outside 0..4 must be refused, not crash
        p[3].ty = 7;
        try {
            compute.upload_particles(p);
            return 4;
        } catch (const particles::Error& e) {
            if (e.code != PCL_ERR_INVALID) return 5;
        }
    } catch (const particles::Error& e) {
        std::fprintf(stderr, "error %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
