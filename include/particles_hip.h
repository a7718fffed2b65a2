/*
 * particles_hip.h — C ABI of the MI355X (gfx950) particle-update library.
 *
 * This is the drop-in boundary for the per-frame particle update that
 * sphereflow/particles runs in src/compute.wgsl behind `pub struct Compute`
 * (reference: src/compute.rs:11-22).  Every entry point is `extern "C"`, takes
 * plain pointers and sizes, returns an int status (0 = ok, negative = error)
 * and never throws or aborts.  A handle owns all of its device memory and is
 * used by one caller at a time (the reference's `&mut self`); different
 * handles are independent.  Host pointers passed in are only read or written
 * during the call.  `pcl_step*` calls are asynchronous on the handle's HIP
 * stream; every call that returns data to the host synchronises that stream.
 *
 * Each declaration cites the reference interface it replaces
 * (file:line relative to the reference checkout).
 */
#ifndef PARTICLES_HIP_H
#define PARTICLES_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ */
/* POD layouts shared with the reference (bit-for-bit).                */
/* ------------------------------------------------------------------ */

/* One particle record, 48 bytes.  src/main.rs:76-83, src/compute.wgsl:1-5. */
typedef struct pcl_particle {
    float    pos[4];  /* xyz + w (the kernel writes w = 1.0)          @0  */
    float    vel[4];  /* xyz + w (the kernel writes w = 1.0)          @16 */
    uint32_t ty;      /* particle type, 0..4                          @32 */
    uint32_t pad[3];  /*                                              @36 */
} pcl_particle;

/* Force-curve polynomial, coeffs[i] multiplies x^i.  src/poly7.rs:6-10,
 * src/compute.wgsl:40-49 (fields h,g,f,e,d,c,b,a = coeffs[0..8]). */
typedef struct pcl_poly7 {
    float coeffs[8];
} pcl_poly7;

/* Mass padded to a 16-byte array stride.  src/main.rs:196-201,
 * src/compute.wgsl:8-10. */
typedef struct pcl_mass_wrap {
    float mass;
    float pad[3];
} pcl_mass_wrap;

#define PCL_NUM_TYPES 5
#define PCL_NUM_POLYS 25

/* The 912-byte uniform block.  src/sim_params.rs:6-17,
 * src/compute.wgsl:17-28. */
typedef struct pcl_sim_params {
    pcl_poly7     attraction_force[PCL_NUM_POLYS];      /* @0   */
    pcl_mass_wrap particle_type_masses[PCL_NUM_TYPES];  /* @800 */
    uint32_t      force_grid_dimensions[3];             /* @880 */
    float         delta_t;                              /* @892 */
    float         max_velocity;                         /* @896 */
    float         bounding_volume_radius;               /* @900 */
    float         cut_off_distance;                     /* @904 */
    float         distance_exponent;                    /* @908 */
} pcl_sim_params;

/* Per-step diagnostics (no counterpart in the reference, which never reads
 * particle state back; requested for the multi-GPU all-reduce). 48 bytes;
 * the same layout is used for the device-resident copy. */
typedef struct pcl_diagnostics {
    double count;           /* number of particles reduced (exact integer)  @0  */
    double kinetic_energy;  /* sum 0.5*m*|v|^2, accumulated in f64          @8  */
    float  pos_max[3];      /* max x, y, z                                  @16 */
    float  neg_pos_min[3];  /* -(min x), -(min y), -(min z): max-reducible  @28 */
    float  pad[2];          /*                                              @40 */
} pcl_diagnostics;

typedef struct pcl_compute pcl_compute; /* opaque handle */

/* Status codes. */
enum {
    PCL_OK               = 0,
    PCL_ERR_INVALID      = -1, /* bad argument (null pointer, size mismatch, ty >= 5, ...) */
    PCL_ERR_HIP          = -2, /* a HIP runtime call failed; see pcl_last_error()          */
    PCL_ERR_NO_DEVICE    = -3, /* no usable gfx950 device                                  */
    PCL_ERR_OUT_OF_RANGE = -4, /* first/count outside the particle or cell range           */
    PCL_ERR_DEVICE       = -5, /* a kernel of the handle flagged a condition only it could see (the order refresh
                                  reserved more slots than its histogram counted; a cursor record read from device
                                  memory had an unknown mode or falloff).  Returned once, by the next call that
                                  synchronises the stream (pcl_synchronize, pcl_read_*, pcl_diagnostics_read,
                                  pcl_debug_cell_indices, pcl_brush_selection); pcl_last_error() says which */
    PCL_ERR_COMM         = -6, /* RCCL could not be loaded, or one of its calls failed                           */
    PCL_ERR_SINGULAR     = -7  /* pcl_poly7_from_points: no polynomial (where the reference returns None)          */
};

/* Which kernel family pcl_step picked (see pcl_last_step_kernel). */
enum {
    PCL_KERNEL_NONE      = 0,
    PCL_KERNEL_STREAM    = 1, /* pair term provably zero: streaming kernel        */
    PCL_KERNEL_PAIRS     = 2  /* general kernel with the all-pairs attraction loop */
};

/* Thread-local description of the last failure on the calling thread. */
const char* pcl_last_error(void);

/* Number of HIP devices visible; negative status on failure. */
int pcl_device_count(void);

/* Default parameter block.  Replaces SimParams::new(), src/sim_params.rs:20-31. */
void pcl_sim_params_default(pcl_sim_params* out);

/*
 * Construct.  Replaces Compute::new(device, particles, force_grid),
 * src/compute.rs:25-145: copies both inputs, parameters start as
 * SimParams::new() (delta_t = 0) exactly as compute.rs:33 does.
 * `n` may be 0.  `n_cells` is not compared with the parameter block's
 * dimensions by the step (the reference never checks either): a cell index
 * past the end is clamped to the last cell.  Only the brush entry points,
 * which need the grid's geometry, require dx*dy*dz == n_cells.
 */
int pcl_create(const pcl_particle* particles, size_t n,
               const float (*force_grid)[4], size_t n_cells,
               int device, pcl_compute** out);

/* Drop.  Replaces the implicit Drop of Compute (wgpu buffers freed). */
int pcl_destroy(pcl_compute* h);

/* Replaces Compute::upload_particles, src/compute.rs:182-196: replaces the
 * whole particle set and N.  The ping-pong position is not reset. */
int pcl_upload_particles(pcl_compute* h, const pcl_particle* particles, size_t n);

/* Structure-of-arrays upload of the same data (host arrays of length n;
 * `ty` one byte per particle). Extension of compute.rs:182 for callers that
 * already hold SoA state. */
int pcl_upload_particles_soa(pcl_compute* h, size_t n,
                             const float* x, const float* y, const float* z,
                             const float* vx, const float* vy, const float* vz,
                             const uint8_t* ty);

/* Fill the particle set on the device with the seeded synthetic scene used
 * by the benchmark (no counterpart in the reference, whose generator is
 * unseeded: src/main.rs:41-48,156-169 give the distribution).  Particle i
 * gets global index first_global_index + i; pos = pos_extent * (u - 0.5),
 * vel = vel_extent * (u - 0.5) per axis with u in [0,1) drawn from a
 * counter-based hash of (seed, global index, axis); ty = global index % 5. */
int pcl_generate_particles(pcl_compute* h, size_t n, uint64_t seed,
                           uint64_t first_global_index,
                           float pos_extent, float vel_extent);

/* Replaces Compute::update_force_grid, src/compute.rs:198-204: whole-field
 * replace from host memory. */
int pcl_update_force_grid(pcl_compute* h, const float (*force_grid)[4], size_t n_cells);

/* Same, from device memory on the handle's device (e.g. the receive buffer
 * of an RCCL broadcast). Ordered on the handle's stream. */
int pcl_update_force_grid_device(pcl_compute* h, const void* dev_force_grid, size_t n_cells);

/* Partial field update for brush edits: cells [first_cell, first_cell+count).
 * Replaces the per-frame whole-field re-upload of src/main.rs:371-380. */
int pcl_patch_force_grid(pcl_compute* h, size_t first_cell,
                         const float (*cells)[4], size_t count);
int pcl_patch_force_grid_device(pcl_compute* h, size_t first_cell,
                                const void* dev_cells, size_t count);

/* Brush edit of a list of cells: field[cell_indices[i]] = cells[i].  Replaces
 * the write loop of Cursor::mouse_moved, src/cursor.rs:84-101 (grid.grid[ix]
 * for ix in modify_vector_indices) followed by the whole-field re-upload of
 * src/main.rs:371-380.  An index listed more than once ends up holding its LAST
 * cell, as after the reference's sequential loop.  The _device form takes both
 * lists from device memory (u32 indices; out-of-range indices are skipped) and
 * is ordered on the stream; there an index listed more than once ends up with
 * one of its cells, which one is not defined. */
int pcl_patch_force_grid_indexed(pcl_compute* h, const uint32_t* cell_indices,
                                 const float (*cells)[4], size_t count);
int pcl_patch_force_grid_indexed_device(pcl_compute* h, const void* dev_cell_indices,
                                        const void* dev_cells, size_t count);

/* ---- field brush -----------------------------------------------------------
 * The editing half of the reference's cursor, run on the device over the
 * resident field instead of on the host followed by a whole-field upload
 * (src/main.rs:371-380).  One 64-byte record carries what Cursor and EditMode
 * hold per event (src/cursor.rs:5-17,144-151). */
enum { PCL_EDIT_CENTERED = 0, PCL_EDIT_SHIFT = 1, PCL_EDIT_ROTATE = 2 };                 /* src/cursor.rs:124-129 */
enum { PCL_FALLOFF_ABRUPT = 0, PCL_FALLOFF_LINEAR = 1, PCL_FALLOFF_INVERSE_DISTANCE = 2 }; /* src/cursor.rs:137-142 */

typedef struct pcl_brush_input {
    float   cursor_pos[3];   /* Cursor::pos                                        */
    float   falloff_dist;    /* EditMode::falloff_dist                             */
    float   cam_right[3];    /* Cursor::rot.x (first column),  src/cursor.rs:168   */
    float   strength;        /* EditMode::strength                                 */
    float   cam_dir[3];      /* Cursor::rot.z (third column),  src/cursor.rs:167   */
    int32_t mode;            /* PCL_EDIT_*                                         */
    int32_t relative;        /* 1 = RelAbE::Relative, 0 = Absolute                 */
    int32_t falloff;         /* PCL_FALLOFF_*                                      */
    int32_t reserved[2];
} pcl_brush_input;

/* Cursor::new + EditMode::default: src/cursor.rs:20-34,206-216. */
void pcl_brush_input_default(pcl_brush_input* out);

/* Replaces Cursor::mouse_down, src/cursor.rs:105-116: selects the cells whose
 * centre (Grid::get_positions, src/grid.rs:119-138, bounds (-r, 2r) and
 * dimensions from the parameter block) lies within falloff_dist of cursor_pos
 * and remembers their vectors and the cursor position.  Asynchronous.  The
 * _device variant reads the record from device memory (e.g. a buffer that was
 * just broadcast to every GPU).  The host variants reject a record whose mode
 * or falloff is not one of the enumerated values with PCL_ERR_INVALID; the
 * _device variants cannot look at the record before the launch: such a record
 * then edits nothing and the next synchronising call returns PCL_ERR_DEVICE. */
int pcl_brush_mouse_down(pcl_compute* h, const pcl_brush_input* in);
int pcl_brush_mouse_down_device(pcl_compute* h, const void* dev_in);

/* Replaces Cursor::mouse_moved, src/cursor.rs:79-103, with
 * EditMode::get_vector, src/cursor.rs:158-203: rewrites exactly the selected
 * cells of the device field.  Does nothing unless a mouse_down is active. */
int pcl_brush_mouse_moved(pcl_compute* h, const pcl_brush_input* in);
int pcl_brush_mouse_moved_device(pcl_compute* h, const void* dev_in);

/* Replaces Cursor::mouse_up, src/cursor.rs:118-121. */
int pcl_brush_mouse_up(pcl_compute* h);

/* The current selection (Cursor::modify_vector_indices, src/cursor.rs:12),
 * ascending; *count receives its size, at most `capacity` indices are
 * written.  Synchronises. */
int pcl_brush_selection(pcl_compute* h, uint32_t* out_indices, size_t capacity, size_t* count);

/* Number of cells of the device field (length of the last whole-field upload). */
size_t pcl_num_cells(const pcl_compute* h);

/* Reads cells [first_cell, first_cell + count) of the device field (the host
 * copy of the reference's Grid is not kept up to date by the device brush). */
int pcl_read_force_grid(pcl_compute* h, float (*out)[4], size_t first_cell, size_t count);

/* Replaces Compute::update_sim_params, src/compute.rs:206-219.  delta_t
 * travels in the block, as in the reference. */
int pcl_update_sim_params(pcl_compute* h, const pcl_sim_params* params);

/* Replaces Compute::compute, src/compute.rs:221-227: one update of every
 * particle from the current buffer into the other one, then flip. */
int pcl_step(pcl_compute* h);

/* n_steps updates, with the same result as n_steps calls of pcl_step (bit for
 * bit).  For callers that do not look at the state, edit the field or change
 * parameters between those steps: while the pair term is inert, runs of
 * consecutive steps are fused into single launches that keep the particles in
 * registers, so n fused steps cost the memory traffic of one (up to 16 per
 * launch; the steps that refresh the particle order stay separate).  No
 * counterpart in the reference, which renders every frame
 * (src/renderer.rs:221-226). */
int pcl_step_n(pcl_compute* h, uint64_t n_steps);

/* Convenience: set delta_t, then pcl_step. */
int pcl_step_dt(pcl_compute* h, float delta_t);

/* The pair term across several handles (one per GPU), each holding a
 * contiguous index range of one global particle set.  The loop of
 * src/compute.wgsl:115-144 runs over ALL particles, so every frame the ranges
 * exchange their positions and types:
 *
 *   pcl_pack_pair_sources_device writes this handle's latest state as 16-byte
 *   records (x, y, z as f32, type as u32) in caller order into device memory
 *   `dev_out`, followed by capacity_records - n padding records (type
 *   0xFFFFFFFF), so equal-sized blocks can be all-gathered; asynchronous on
 *   the handle's stream.
 *
 *   pcl_step_pairs_gathered is pcl_step with the partners taken from
 *   `dev_sources`: n_sources such records, the blocks of all ranges
 *   concatenated in global index order (padding is skipped).  Each particle
 *   adds its terms in that order, which is the order of a single handle
 *   holding the whole set, so the sharded result equals the unsharded one bit
 *   for bit.  Always runs the exact all-pairs kernel.  The buffer must stay
 *   valid until the step has run on the handle's stream.
 *
 * No counterpart in the reference (one device there). */
int pcl_pack_pair_sources_device(pcl_compute* h, void* dev_out, size_t capacity_records);
int pcl_step_pairs_gathered(pcl_compute* h, const void* dev_sources, size_t n_sources);

/* Force the general (pair-loop) kernel even when the pair term is provably
 * zero; for equivalence tests.  0 = automatic choice (default). */
int pcl_force_pair_kernel(pcl_compute* h, int enable);

/* How the pair term (src/compute.wgsl:115-144) is evaluated when it is not
 * provably zero.  PCL_PAIRS_ALL: the shader's sweep over all N particles in
 * index order (bit-exact summation order, reproducible).  PCL_PAIRS_BINNED:
 * particles are sorted into cells of edge >= cut_off_distance (or half of it)
 * and only the cells within reach are swept: same skip tests and force
 * expression, O(N k) instead of O(N^2); a particle's terms are added cell by
 * cell, within a cell in ascending caller index (the sort is a stable radix
 * sort of (cell, caller index)), so results are reproducible from run to run
 * and independent of the memory order before the step, and equal the exact
 * sweep's up to fp32 summation order.  At most 20 cells per axis, 64 from 2^18
 * particles on.
 * PCL_PAIRS_AUTO (default): binned from 2^15 particles on when the cut-off
 * allows at least 3 cells per axis, the exact sweep otherwise.
 * pcl_step_pairs_gathered (several handles) always runs the exact sweep, so a
 * sharded run equals a one-handle run bit for bit only where that one also
 * runs the exact sweep: under PCL_PAIRS_ALL, or below 2^15 particles. */
enum { PCL_PAIRS_AUTO = 0, PCL_PAIRS_ALL = 1, PCL_PAIRS_BINNED = 2 };
int pcl_set_pair_mode(pcl_compute* h, int mode);

/* PCL_KERNEL_* of the most recent step. */
int pcl_last_step_kernel(const pcl_compute* h);

/* When enabled, every step also reduces diagnostics of the state it writes
 * (fused into the step kernel, no extra pass over HBM). */
int pcl_enable_step_diagnostics(pcl_compute* h, int enable);

/* Replaces the public field Compute::num_particles, src/compute.rs:19. */
size_t pcl_num_particles(const pcl_compute* h);

/* Number of steps taken since creation. */
uint64_t pcl_step_count(const pcl_compute* h);

/* Read back AoS records of the latest state, caller order
 * (the reference exposes particles_buffers, src/compute.rs:13, stride 48,
 * pos @0, ty @32: src/main.rs:86-107).  pos.w and vel.w read back as 1.0. */
int pcl_read_particles(pcl_compute* h, pcl_particle* out, size_t first, size_t count);

/* Read back SoA arrays of the latest state (any pointer may be NULL). */
int pcl_read_soa(pcl_compute* h, size_t first, size_t count,
                 float* x, float* y, float* z,
                 float* vx, float* vy, float* vz, uint8_t* ty);

/* Pack the latest state into a device buffer of 48-byte records, the
 * renderer's instance layout (src/main.rs:86-107, src/renderer.rs:253-257). */
int pcl_pack_particles_device(pcl_compute* h, void* dev_out, size_t first, size_t count);

/* All n records in whatever order the particles currently have in memory
 * (every particle exactly once).  An instance stream does not care about the
 * order of its instances (src/renderer.rs:253-257 draws them all), and this
 * is a straight streaming pass even while the particles are kept in field-tile
 * order.  dev_out must hold n records. */
int pcl_pack_particles_device_unordered(pcl_compute* h, void* dev_out);

/* Diagnostics of the latest state, synchronous.  Uses the fused result when
 * step diagnostics are enabled and current, otherwise runs a reduction. */
int pcl_diagnostics_read(pcl_compute* h, pcl_diagnostics* out);

/* Same, written asynchronously to `dev_out` (48 bytes of device memory),
 * ordered on the handle's stream: the input of the per-step all-reduce. */
int pcl_diagnostics_device(pcl_compute* h, void* dev_out);

/* pcl_diagnostics_device for callers that cannot allocate: code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc.  No
 * counterpart in the reference.  Calls no hipMalloc / hipFree and nothing that
 * allocates: the reduction's block partials go to a file-scope device array of
 * the library (196,608 bytes per GPU, held from load until the process ends)
 * and the 48-byte record is written straight to `dev_out` (8-byte aligned
 * device memory), asynchronously on the handle's stream.  Always reduces the
 * latest state (never reuses a fused per-step result), so on the same state it
 * returns what pcl_diagnostics_device returns when step diagnostics are off.
 * At most PCL_NOALLOC_MAX_PARTICLES particles (PCL_ERR_INVALID beyond).  All
 * calls on one GPU share that GPU's array: calls that overlap on one GPU (two
 * handles on different streams) race; calls on one stream are ordered. */
#define PCL_NOALLOC_MAX_PARTICLES (1ull << 32)
int pcl_diagnostics_device_noalloc(pcl_compute* h, void* dev_out);

/* Flat field-cell index of every particle's current position
 * (src/compute.wgsl:81-89,147), for bit-exact index tests. */
int pcl_debug_cell_indices(pcl_compute* h, uint32_t* out, size_t first, size_t count);

/* Test hook for the bin work of the cell-binned pair term (no counterpart in the
 * reference, whose loop of src/compute.wgsl:115-144 visits every particle): runs
 * the sort by cut-off cell that the binned step runs before its sweep and returns,
 * in caller order, the cell key of every particle ((cx * n + cy) * n + cz for n
 * cells per axis) and, when bin_end is not null, for every cell one past the last
 * slot of the sorted order that holds one of its particles (n^3 entries;
 * bin_capacity is the room in bin_end).  The particles are left in cell order;
 * results of later steps and reads are unaffected. */
int pcl_debug_pair_cells(pcl_compute* h, uint32_t* cell_of_particle, uint32_t* bin_end, size_t bin_capacity,
                         uint32_t* cells_per_axis, uint32_t* reach);

/* Test hook: the caller's index of the particle in every memory slot (the identity
 * until the library has reordered the set: field-tile order of the streaming step,
 * cell order of the binned pair term).  No counterpart in the reference, which
 * never reorders (src/compute.wgsl processes particles in buffer order). */
int pcl_debug_slot_order(pcl_compute* h, uint32_t* caller_index_of_slot, size_t capacity);

/* Test hook: a checksum of the whole current state that does not depend on the
 * order of the particles in memory.  Particle i (caller order) contributes
 *   h = (first_global_index + i + 1) * 0x9E3779B97F4A7C15;  for w in (x, y, z, vx, vy, vz bits, type):
 *   h = (h ^ w) * 0xFF51AFD7ED558CCD; h ^= h >> 32          (all mod 2^64)
 * and *sum / *xor_of_all receive the sum and the xor of the contributions: two
 * handles agree on both iff, up to collisions, every particle has the same bits
 * in both.  For comparing states too large to read back (a whole 2^28-particle
 * run on two different kernel paths); with first_global_index = the first index
 * of a handle's range, the sums of the ranges add up (mod 2^64) and their xors
 * xor to the words of one handle holding the whole set.  No counterpart in the
 * reference. */
int pcl_debug_state_checksum(pcl_compute* h, uint64_t first_global_index, uint64_t* sum, uint64_t* xor_of_all);

/* Run subsequent work of this handle on `hip_stream` (a hipStream_t owned by
 * the caller, e.g. torch's current stream); NULL restores the internal one. */
int pcl_set_stream(pcl_compute* h, void* hip_stream);

/* Run subsequent work of this handle on the device's null (legacy default)
 * stream.  That is the stream PyTorch and its RCCL collectives order against
 * unless told otherwise (its handle value is 0, which pcl_set_stream takes as
 * "restore the internal stream"); the handle's internal stream is
 * non-blocking and is NOT ordered with it. */
int pcl_use_null_stream(pcl_compute* h);

/* Block until the handle's stream is idle. */
int pcl_synchronize(pcl_compute* h);

/* HIP-event stopwatch on the handle's stream: start records an event, stop
 * records another, waits for it and returns the elapsed milliseconds. */
int pcl_timer_start(pcl_compute* h);
int pcl_timer_stop(pcl_compute* h, float* elapsed_ms);

/* Memory order of the particles (no counterpart in the reference, which
 * processes particles in buffer order).  While the pair term is inert the
 * library may keep the particle arrays sorted by the field tile each particle
 * sits in, refreshed every `interval` steps by one counting-sort pass, so that
 * the per-particle field fetch stays cache-resident for fields larger than an
 * XCD's L2.  Per-particle results are unchanged bit for bit and every read
 * returns caller order.  interval: -1 automatic (default: every 24 steps for
 * fields above 16 KiB and at least 2^20 particles, except where the field is
 * staged in LDS: up to 64 KiB with 2^22 to 2^24 particles), 0 never, k > 0 every k steps. */
int pcl_set_reorder_interval(pcl_compute* h, int interval);

/* The interval currently in force: 0 when the order is not maintained (small
 * sets, fields small enough for LDS, or interval 0); in automatic mode it
 * starts at 24 and is paced by how far the particles drift between refreshes
 * (a counting step reports how many tiles each block's particles fall into:
 * up to 3 doubles the interval, at most to 96; 8 or more halves it again). */
int pcl_reorder_interval_in_use(const pcl_compute* h);

/* Sort by field tile now (also the first step does when reordering is on). */
int pcl_reorder_now(pcl_compute* h);

/* Number of sort passes run so far. */
uint64_t pcl_reorder_count(const pcl_compute* h);

/* What one pcl_step launched (pcl_step_timing_read). */
enum {
    PCL_STEP_NONE           = 0,
    PCL_STEP_STREAM         = 1, /* step_stream_kernel                                      */
    PCL_STEP_PAIRS          = 2, /* step_pairs_kernel                                       */
    PCL_STEP_STREAM_COUNT   = 3, /* step_stream_kernel that also histograms the field tiles */
    PCL_STEP_STREAM_SCATTER = 4, /* step_scatter_kernel: update + refresh of the tile order */
    PCL_STEP_PAIRS_BINNED   = 5, /* sort by cut-off cell + step_pairs_binned_kernel            */
    PCL_STEP_PAIRS_GATHERED = 6, /* step_pairs_kernel over partner records given by the caller */
    PCL_STEP_STREAM_FUSED   = 7  /* step_stream_fused_kernel: several updates in one launch      */
};

/* Per-launch timing: between begin and read every pcl_step is timed by a HIP
 * event pair on the handle's stream (no synchronisation until the read): a step
 * that is one plain kernel launch hands the pair to that launch, which stamps
 * it with the kernel's own begin and end and puts nothing else on the stream;
 * every other kind of step lies between two event records.
 * read waits for the last step, fills elapsed_ms[i] / kinds[i] (PCL_STEP_*) for
 * the recorded steps and stops the recording. */
int pcl_step_timing_begin(pcl_compute* h, size_t max_steps);
int pcl_step_timing_read(pcl_compute* h, float* elapsed_ms, int32_t* kinds,
                         size_t capacity, size_t* count);

/* PCI bus id of the handle's GPU ("0000:c1:00.0"), for reports that must show
 * which physical devices the ranks of a job ran on.  capacity >= 16. */
int pcl_device_pci_bus_id(const pcl_compute* h, char* out, size_t capacity);

/* Geometry of the streaming kernel's launch (for reports). */
int pcl_launch_geometry(const pcl_compute* h, uint32_t* grid_blocks,
                        uint32_t* block_threads, uint32_t* particles_per_lane);

/* Residency of the plain streaming step at its last launch (no counterpart in the
 * reference).  The library limits how many blocks of that kernel are resident per
 * CU by asking for dynamic LDS the kernel never touches (the memory system
 * delivers more with fewer waves waiting on it); *intended is the limit the
 * launch shape asked for (0: none), *granted what the HIP occupancy calculation
 * reports for the kernel with that much LDS (with no limit: whatever fits).  When
 * the two differ the limit is not used, and the mismatch is reported once on
 * stderr and in pcl_last_error.  Both 0 before the first such launch. */
int pcl_launch_residency(const pcl_compute* h, uint32_t* intended_blocks_per_cu,
                         uint32_t* granted_blocks_per_cu);

/* The cell-binned pair term (pcl_set_pair_mode) has two sweeps with the same
 * results: one particle per lane with its own candidate ranges, or one wave per
 * 64 slots with the candidates shared through LDS, which wins where the cut-off
 * cells are crowded.  The library picks by the particles per occupied cell that
 * the cell sort of two steps earlier counted.  *shared_sweep: 1 when the last
 * binned step ran the shared sweep; *cells_occupied: the latest count read (0:
 * none yet).  No counterpart in the reference (src/compute.wgsl:115-144 visits
 * every pair). */
int pcl_pair_sweep_info(const pcl_compute* h, uint32_t* shared_sweep, uint32_t* cells_occupied);

/* Tuning knobs for measurements (0 keeps the built-in defaults; results never
 * depend on them).  grid_blocks caps the grid of the streaming kernel (the
 * blocks then stride over the tiles).  variant is a bit set:
 *   0x1      plain instead of non-temporal access to the particle arrays
 *   0x2      never stage the field in LDS
 *   0x8      separate counting pass instead of the counting step
 *   0xF0     log2 of the cells per sort tile (automatic: 8 to 10, i.e. 16 x 16 to 32 x 32 cells of a flat field, more the
 *            fewer particles a cell holds)
 *   0xF000   point of the refresh interval the tile order aims at, in tenths (default 5)
 *   0x10000  cut-off cells of the binned pair term: one-cell reach only
 *   0x20000  cut-off cells: at most 20 per axis
 *   0x400000 the streaming kernels fetch the 16-byte cells instead of the copy packed to 12 bytes
 *   0x80000  the streaming step updates the particle arrays in place instead of ping-pong
 *   0xF000000  resident blocks per CU of the plain streaming step (1-12; 15 = as many as fit), and
 *   0x70000000 its block size (0: 512 threads, 1: 256, 2: 128, 3: 64); both zero = automatic
 *            (256 threads, 3 blocks per CU from 2^23 particles on; with fused diagnostics 4, or 512 threads x 3 when the
 *            field exceeds 4 MiB; 512 threads, as many as fit below)
 *   0x300000 resident blocks per CU of the counting step (1-3; 0 = automatic)
 *   0x4      the scalar path of the streaming kernels only (full divisions) instead of the packed one
 *   0x100    plain streaming step: runs of 128 tiles dealt so that each XCD takes 16 consecutive ones (automatic
 *            from 3 * 2^23 to 2^26 particles in tile order with at most 40 per field cell), 0x400: never
 *   0x800000 / 0x80000000  cell-binned pair term: always the sweep with one particle per lane / always the sweep with
 *            the candidates shared by a wave (automatic: by the particles per occupied cut-off cell)
 *   0x200    particles in caller order are sorted once instead of twice before their first step
 *   0x800    step timing (pcl_step_timing_begin): an event record on either side of EVERY step instead of
 *            handing the two events of a one-kernel step to that launch
 *   0x40000  tests only: the next refresh of the particle order scatters with other keys than its
 *            histogram counted, which the kernel reports as PCL_ERR_DEVICE (the state is then unusable) */
int pcl_set_tuning(pcl_compute* h, uint32_t grid_blocks, uint32_t variant);

/* ---- authoring helpers (host only, no device needed) -----------------------
 * What the reference's GUI uses to build the update's inputs; fp32 operation
 * for operation in the reference's order. */

/* Poly7::from_points, src/poly7.rs:37-58: the polynomial through eight (x, y)
 * points, by inverting the 8x8 Vandermonde matrix with the reference's
 * Gauss-Jordan elimination without pivot search (src/poly7.rs:120-147) and
 * multiplying by the y values.  x^j is f32::powi (square-and-multiply, as
 * compiler-rt's __powisf2).  PCL_ERR_SINGULAR where the reference returns None
 * (a non-finite entry in the inverse, e.g. a repeated x). */
int pcl_poly7_from_points(const float (*points)[2], pcl_poly7* out);
/* Poly7::eval, src/poly7.rs:60-68 (the GUI's plotting order, ascending powers
 * with a running product; the shader evaluates in another order). */
float pcl_poly7_eval(const pcl_poly7* p, float x);
/* Grid::new_centered + get_force_vectors, src/grid.rs:68-103: cell (ix,iy,iz)
 * holds (nx,ny,nz)*0.5 - (ix,iy,iz), w = 1; x slowest, z fastest. */
void pcl_grid_centered(size_t nx, size_t ny, size_t nz, float (*out)[4]);
/* Grid::new_uniform + get_force_vectors, src/grid.rs:49-64,97-103. */
void pcl_grid_uniform(size_t nx, size_t ny, size_t nz, const float value[3], float (*out)[4]);
/* Grid::get_positions, src/grid.rs:119-138: cell-centre positions for bounds
 * (pos, dir). */
void pcl_grid_positions(size_t nx, size_t ny, size_t nz, const float bounds_pos[3], const float bounds_dir[3],
                        float (*out)[4]);
/* SimParams::new_force_grid_centered (centered != 0) / new_force_grid_zero,
 * src/sim_params.rs:33-60, as upload-ready cells for the block's dimensions. */
int pcl_sim_params_force_grid(const pcl_sim_params* params, int centered, float (*out)[4], size_t capacity_cells);

/* ======================================================================
 * Several GPUs of one node (RCCL over xGMI inside the library).
 *
 * No counterpart in the reference, which runs on one device: a host would
 * change its construction (src/main.rs:232-236) and its per-frame calls
 * (src/main.rs:371-390, src/renderer.rs:225).  Particles shard by contiguous
 * index range, rank g of G holding [g*N/G, (g+1)*N/G); the field and the
 * parameter block are replicated.  With the pair term inert
 * (src/compute.wgsl:115-144 adds nothing) a step involves no collective on
 * the data path.  What travels: the field (or its dirty range, or a brush's
 * 64-byte cursor record) by ncclBroadcast into every replica; with the pair
 * term active, one ncclAllGather of 16 bytes per particle per step; with step
 * diagnostics on, one ncclGroupStart/End pair of all-reduces of the 48-byte
 * record per step.  Every collective runs on the handle's own stream, ordered
 * with its kernels, except that per-step all-reduce: nothing in the next step
 * needs its result, so it runs on a stream of the communicator's own, beside
 * the following step (the ranks then wait for one another at most two steps
 * apart instead of after every step).  RCCL is loaded at first use (dlopen; PCL_RCCL_LIBRARY
 * names the file when the default search is not wanted); without it these
 * calls return PCL_ERR_COMM and the rest of the library works.
 *
 * Two ways to form the communicator:
 *   pcl_comm_*  : one process (or thread) per GPU.  Rank 0 makes a unique id,
 *                 the host carries its 128 bytes to the other ranks by any
 *                 means, every rank attaches its own handle.  The pcl_comm_*
 *                 calls below are collective: every rank makes the same call,
 *                 in the same order, with the same sizes and root; pointers
 *                 to host data are only looked at on the root.
 *   pcl_group_* : one process holds a handle per GPU (ncclCommInitAll) and
 *                 every call works on all of them, rank 0 being the root.
 * ====================================================================== */

#define PCL_UNIQUE_ID_BYTES 128
int pcl_comm_unique_id(void* out_id /* PCL_UNIQUE_ID_BYTES */);
int pcl_comm_attach(pcl_compute* h, const void* unique_id, int rank, int world);
int pcl_comm_detach(pcl_compute* h);          /* pcl_destroy does it too */
int pcl_comm_rank(const pcl_compute* h);      /* -1 without a communicator */
int pcl_comm_world(const pcl_compute* h);     /*  0 without a communicator */
/* The rank count as the communicator itself reports it (ncclCommCount): what RCCL
 * saw, as opposed to what pcl_comm_attach was told; 0 without a communicator,
 * -1 when the query fails. */
int pcl_comm_count(const pcl_compute* h);

/* Compute::update_force_grid (src/compute.rs:198-204) on every replica: the
 * root's cells are broadcast straight into each handle's field. */
int pcl_comm_update_force_grid(pcl_compute* h, const float (*force_grid)[4], size_t n_cells, int root);
/* Brush edits (src/main.rs:371-380 uploads the whole field instead): only the
 * dirty range, or only the listed cells (20 bytes each), travel. */
int pcl_comm_patch_force_grid(pcl_compute* h, size_t first_cell, const float (*cells)[4], size_t count, int root);
int pcl_comm_patch_force_grid_indexed(pcl_compute* h, const uint32_t* cell_indices, const float (*cells)[4],
                                      size_t count, int root);
/* Cursor::mouse_down / mouse_moved (src/cursor.rs:79-116) on every replica:
 * the root's 64-byte record is broadcast and each GPU edits its own field. */
int pcl_comm_brush_mouse_down(pcl_compute* h, const pcl_brush_input* in, int root);
int pcl_comm_brush_mouse_moved(pcl_compute* h, const pcl_brush_input* in, int root);
/* Compute::compute (src/compute.rs:221-227) of this rank's range.  Pair term
 * active: all ranks exchange positions and types first (all-gather) and every
 * particle sums over all of them in global index order, which gives the bits
 * of one handle holding every particle under PCL_PAIRS_ALL.  Step diagnostics
 * enabled: also starts the all-reduce of the step's record, beside the steps
 * that follow (the result is what pcl_comm_diagnostics_device points at; the
 * calls that wait for the handle's stream wait for it too; steps of such a
 * handle should all go through this call).  No host synchronisation,
 * except in the first step after pcl_comm_attach, after a call that replaced
 * the particle set (pcl_upload_particles[_soa], pcl_generate_particles) and
 * after a change of what a step exchanges (pair term active or inert, step
 * diagnostics on or off): that step first agrees with the other ranks on the
 * padded range length and on these two settings (one all-reduce, read back on
 * the host) and returns PCL_ERR_INVALID on every rank when the settings
 * differ.  All ranks must therefore replace their particle sets and change
 * these settings between the same two steps. */
int pcl_comm_step(pcl_compute* h);
/* Diagnostics of the latest state of ALL ranges, identical on every rank:
 * read back (a collective of its own, synchronises), or the device address of
 * the record the latest pcl_comm_step with step diagnostics all-reduced, valid
 * for work enqueued on the handle's stream AFTER this call (the call orders
 * that stream behind the all-reduce; no host synchronisation). */
int pcl_comm_diagnostics_read(pcl_compute* h, pcl_diagnostics* out);
int pcl_comm_diagnostics_device(pcl_compute* h, const void** dev_record);

typedef struct pcl_group pcl_group; /* one handle per GPU, driven by one process */

/* Compute::new (src/compute.rs:25) over `n_devices` GPUs: particle i goes to
 * the handle whose range holds i, the field to all of them. */
int pcl_group_create(const int* devices, int n_devices, const pcl_particle* particles, size_t n,
                     const float (*force_grid)[4], size_t n_cells, pcl_group** out);
int pcl_group_destroy(pcl_group* g);
int pcl_group_size(const pcl_group* g);
pcl_compute* pcl_group_handle(pcl_group* g, int i);   /* for per-GPU settings and reads; owned by the group */
size_t pcl_group_num_particles(const pcl_group* g);   /* the whole set */
int pcl_group_upload_particles(pcl_group* g, const pcl_particle* particles, size_t n);
/* pcl_generate_particles for the whole set: every GPU fills its own range. */
int pcl_group_generate_particles(pcl_group* g, size_t n_total, uint64_t seed, float pos_extent, float vel_extent);
int pcl_group_update_force_grid(pcl_group* g, const float (*force_grid)[4], size_t n_cells);
int pcl_group_patch_force_grid(pcl_group* g, size_t first_cell, const float (*cells)[4], size_t count);
int pcl_group_patch_force_grid_indexed(pcl_group* g, const uint32_t* cell_indices, const float (*cells)[4], size_t count);
int pcl_group_update_sim_params(pcl_group* g, const pcl_sim_params* params);
int pcl_group_enable_step_diagnostics(pcl_group* g, int enable);
int pcl_group_step(pcl_group* g);
int pcl_group_brush_mouse_down(pcl_group* g, const pcl_brush_input* in);
int pcl_group_brush_mouse_moved(pcl_group* g, const pcl_brush_input* in);
int pcl_group_brush_mouse_up(pcl_group* g);
int pcl_group_diagnostics_read(pcl_group* g, pcl_diagnostics* out);
/* Records [first, first + count) of the whole set, caller order. */
int pcl_group_read_particles(pcl_group* g, pcl_particle* out, size_t first, size_t count);
/* Cells of one replica of the field (they are all equal). */
int pcl_group_read_force_grid(pcl_group* g, int replica, float (*out)[4], size_t first_cell, size_t count);
int pcl_group_synchronize(pcl_group* g);

#ifdef __cplusplus
} /* extern "C" */
#endif

#endif /* PARTICLES_HIP_H */
