/* The reference's default scene through the C ABI, in plain C: the 1000-particle
 * lattice of src/main.rs:117-154 in the centred 10 x 10 x 10 field of
 * src/grid.rs:68-95, stepped at a fixed dt the way App::update / Renderer::render
 * drive Compute (src/main.rs:371-390, src/renderer.rs:221-226), then read back.
 *
 *   gcc -std=c11 -Iinclude examples/default_scene.c -Lparticles_amd -lparticles_hip \
 *       -Wl,-rpath,$PWD/particles_amd -Wl,-rpath-link,/opt/rocm/lib -o default_scene
 */
#include "particles_hip.h"

#include <stdio.h>
#include <stdlib.h>

#define CHECK(call)                                                         \
    do {                                                                    \
        int rc_ = (call);                                                   \
        if (rc_ != PCL_OK) {                                                \
            fprintf(stderr, "%s: %d (%s)\n", #call, rc_, pcl_last_error()); \
            return 1;                                                       \
        }                                                                   \
    } while (0)

int main(int argc, char** argv)
{
    const int steps = argc > 1 ? atoi(argv[1]) : 16;
    enum { NX = 10, NY = 10, NZ = 10, N = NX * NY * NZ };

    /* src/main.rs:117-154: pos = (ix/nx*5, iy/ny*2, iz/nz*2, 1), vel = 0, ty = flat index % 5 */
    static pcl_particle particles[N];
    /* src/grid.rs:68-95: v = (nx, ny, nz)/2 - (ix, iy, iz); x slowest, z fastest */
    static float field[N][4];
    for (int ix = 0, i = 0; ix < NX; ++ix)
        for (int iy = 0; iy < NY; ++iy)
            for (int iz = 0; iz < NZ; ++iz, ++i) {
                particles[i].pos[0] = (float)ix / (float)NX * 5.0f;
                particles[i].pos[1] = (float)iy / (float)NY * 2.0f;
                particles[i].pos[2] = (float)iz / (float)NZ * 2.0f;
                particles[i].pos[3] = 1.0f;
                particles[i].ty = (uint32_t)(i % 5);
                field[i][0] = (float)NX / 2.0f - (float)ix;
                field[i][1] = (float)NY / 2.0f - (float)iy;
                field[i][2] = (float)NZ / 2.0f - (float)iz;
                field[i][3] = 1.0f;
            }

    pcl_compute* compute = NULL;
    /* (ISO C wants the cast to a pointer to const rows; C++ converts by itself) */
    CHECK(pcl_create(particles, N, (const float (*)[4])field, N, 0, &compute));   /* Compute::new */

    pcl_sim_params params;
    pcl_sim_params_default(&params);                               /* SimParams::new */
    params.delta_t = 1.0f / 60.0f;                                 /* the frame time travels in the block */
    CHECK(pcl_update_sim_params(compute, &params));                /* Compute::update_sim_params */

    for (int s = 0; s < steps; ++s) CHECK(pcl_step(compute));      /* Compute::compute, asynchronous */

    CHECK(pcl_read_particles(compute, particles, 0, N));           /* latest state, caller order */
    pcl_diagnostics d;
    CHECK(pcl_diagnostics_read(compute, &d));
    printf("steps %d particles %.0f kinetic_energy %.9g\n", steps, d.count, d.kinetic_energy);
    printf("particle 0: pos %.9g %.9g %.9g vel %.9g %.9g %.9g\n", (double)particles[0].pos[0],
           (double)particles[0].pos[1], (double)particles[0].pos[2], (double)particles[0].vel[0],
           (double)particles[0].vel[1], (double)particles[0].vel[2]);
    printf("particle 999: pos %.9g %.9g %.9g vel %.9g %.9g %.9g\n", (double)particles[999].pos[0],
           (double)particles[999].pos[1], (double)particles[999].pos[2], (double)particles[999].vel[0],
           (double)particles[999].vel[1], (double)particles[999].vel[2]);
    CHECK(pcl_destroy(compute));
    return 0;
}
