/* The update over every GPU of the node from ONE process, in plain C: a
 * pcl_group shards the particles by index range and keeps the field replicas
 * equal with RCCL inside the library.  The program walks through every
 * collective there is — field broadcast at construction, cursor-record
 * broadcast of a brush stroke (src/cursor.rs:79-116), dirty-range and indexed
 * patches (src/main.rs:371-380 uploads the whole field instead), the
 * diagnostics all-reduce per step, and the position/type all-gather of the
 * pair term (src/compute.wgsl:115-144) — and writes what it read back to a
 * file, so that a test can compare it with a CPU model of one device.
 *
 *   gcc -std=c11 -Iinclude examples/group_scene.c -Lparticles_amd -lparticles_hip \
 *       -Wl,-rpath,$PWD/particles_amd -Wl,-rpath-link,/opt/rocm/lib -o group_scene
 *   ./group_scene out.bin [n_devices [device]]
 * With `device` every rank of the group sits on that one GPU, which RCCL refuses; the tests use it
 * together with a host-staged stand-in for RCCL (PCL_RCCL_LIBRARY) to run several ranks on one GPU.
 */
#include "particles_hip.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(call)                                                         \
    do {                                                                    \
        int rc_ = (call);                                                   \
        if (rc_ != PCL_OK) {                                                \
            fprintf(stderr, "%s: %d (%s)\n", #call, rc_, pcl_last_error()); \
            return 1;                                                       \
        }                                                                   \
    } while (0)

enum { NX = 24, NY = 16, NZ = 2, CELLS = NX * NY * NZ, N = 3001, N_PAIRS = 601 };

/* a small deterministic generator, so that the test can build the same inputs */
static unsigned int lcg(unsigned int* s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }
static float unit(unsigned int* s) { return (float)lcg(s) * (1.0f / 16777216.0f); }

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s out.bin [n_devices [device]]\n", argv[0]); return 2; }
    int n_devices = argc > 2 ? atoi(argv[2]) : pcl_device_count();
    if (n_devices < 1) { fprintf(stderr, "no device: %s\n", pcl_last_error()); return 1; }
    int devices[64];
    if (n_devices > 64) n_devices = 64;
    for (int i = 0; i < n_devices; ++i) devices[i] = argc > 3 ? atoi(argv[3]) : i;

    static pcl_particle particles[N];
    static float field[CELLS][4];
    unsigned int seed = 2024u;
    for (int i = 0; i < N; ++i) {
        for (int c = 0; c < 3; ++c) {
            particles[i].pos[c] = 20.0f * (unit(&seed) - 0.5f);
            particles[i].vel[c] = 10.0f * (unit(&seed) - 0.5f);
        }
        particles[i].pos[3] = particles[i].vel[3] = 1.0f;
        particles[i].ty = (uint32_t)(i % 5);
    }
    for (int i = 0; i < CELLS; ++i) {
        for (int c = 0; c < 3; ++c) field[i][c] = 2.0f * (unit(&seed) - 0.5f);
        field[i][3] = 1.0f;
    }

    pcl_group* g = NULL;
    CHECK(pcl_group_create(devices, n_devices, particles, N, (const float (*)[4])field, CELLS, &g));   /* field broadcast */

    pcl_sim_params params;
    pcl_sim_params_default(&params);
    params.delta_t = 1.0f / 60.0f;
    params.force_grid_dimensions[0] = NX; params.force_grid_dimensions[1] = NY; params.force_grid_dimensions[2] = NZ;
    const float masses[5] = {1.0f, 0.5f, 2.0f, 1.5f, 0.75f};
    for (int k = 0; k < 5; ++k) params.particle_type_masses[k].mass = masses[k];
    CHECK(pcl_group_update_sim_params(g, &params));
    CHECK(pcl_group_enable_step_diagnostics(g, 1));                /* all-reduce of the 48-byte record per step */
    for (int s = 0; s < 3; ++s) CHECK(pcl_group_step(g));

    pcl_brush_input brush;                                         /* a stroke: only 64 bytes travel per event */
    pcl_brush_input_default(&brush);
    brush.mode = PCL_EDIT_ROTATE; brush.relative = 0; brush.falloff = PCL_FALLOFF_LINEAR;
    brush.falloff_dist = 4.0f; brush.strength = 1.5f;
    brush.cursor_pos[0] = 1.0f; brush.cursor_pos[1] = -2.0f; brush.cursor_pos[2] = 0.5f;
    CHECK(pcl_group_brush_mouse_down(g, &brush));
    brush.cursor_pos[0] = 1.5f; brush.cursor_pos[1] = -1.0f;
    CHECK(pcl_group_brush_mouse_moved(g, &brush));
    CHECK(pcl_group_brush_mouse_up(g));

    static float patch[40][4];
    for (int i = 0; i < 40; ++i) for (int c = 0; c < 4; ++c) patch[i][c] = (float)(i * 4 + c) / 16.0f - 5.0f;
    CHECK(pcl_group_patch_force_grid(g, 100, (const float (*)[4])patch, 40));          /* dirty range */
    const uint32_t listed[3] = {CELLS - 1, 0, 333};
    CHECK(pcl_group_patch_force_grid_indexed(g, listed, (const float (*)[4])patch, 3)); /* listed cells */
    for (int s = 0; s < 4; ++s) CHECK(pcl_group_step(g));

    static pcl_particle stream_out[N];
    static float field_out[CELLS][4];
    pcl_diagnostics stream_diag;
    CHECK(pcl_group_read_particles(g, stream_out, 0, N));
    CHECK(pcl_group_diagnostics_read(g, &stream_diag));
    CHECK(pcl_group_read_force_grid(g, n_devices - 1, field_out, 0, CELLS));            /* the last replica */

    /* the pair term: every step all-gathers 16 bytes per particle */
    CHECK(pcl_group_upload_particles(g, particles, N_PAIRS));
    params.cut_off_distance = 6.0f;
    for (int k = 0; k < PCL_NUM_POLYS; ++k)
        for (int c = 0; c < 3; ++c) params.attraction_force[k].coeffs[c] = 0.05f * (unit(&seed) - 0.5f);
    CHECK(pcl_group_update_sim_params(g, &params));
    for (int i = 0; i < n_devices; ++i) CHECK(pcl_set_pair_mode(pcl_group_handle(g, i), PCL_PAIRS_ALL));
    for (int s = 0; s < 3; ++s) CHECK(pcl_group_step(g));
    static pcl_particle pairs_out[N_PAIRS];
    pcl_diagnostics pairs_diag;
    CHECK(pcl_group_read_particles(g, pairs_out, 0, N_PAIRS));
    CHECK(pcl_group_diagnostics_read(g, &pairs_diag));

    FILE* f = fopen(argv[1], "wb");
    if (!f) { perror(argv[1]); return 1; }
    fwrite(&params, sizeof params, 1, f);               /* the final parameter block (pair curves included) */
    fwrite(stream_out, sizeof stream_out, 1, f);
    fwrite(field_out, sizeof field_out, 1, f);
    fwrite(&stream_diag, sizeof stream_diag, 1, f);
    fwrite(pairs_out, sizeof pairs_out, 1, f);
    fwrite(&pairs_diag, sizeof pairs_diag, 1, f);
    fclose(f);
    printf("group of %d GPU(s): %zu particles, kinetic energy %.9g after the streaming steps, %.9g after the pair steps\n",
           pcl_group_size(g), (size_t)stream_diag.count, stream_diag.kinetic_energy, pairs_diag.kinetic_energy);
    CHECK(pcl_group_destroy(g));
    return 0;
}
