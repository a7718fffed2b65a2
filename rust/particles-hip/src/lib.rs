//! Drop-in for `compute::Compute` of sphereflow/particles (src/compute.rs)
//! backed by the gfx950 HIP library through its C ABI
//! (include/particles_hip.h).
//!
//! NOT COMPILED OR TESTED in this repository's build environment (no Rust
//! toolchain there); the tested boundary is the C ABI itself.
//!
//! Same surface as the reference: `Compute::new`, `upload_particles`,
//! `update_force_grid`, `update_sim_params`, `compute`, `num_particles`.
//! The wgpu `&Device` / `&mut ComputePass` arguments disappear (the HIP device
//! ordinal is fixed at construction); `Particle` and `SimParams` are the
//! reference's own `#[repr(C)]` types, passed through as bytes.
use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
struct PclCompute {
    _private: [u8; 0],
}

#[repr(C)]
struct PclGroup {
    _private: [u8; 0],
}

/// `pcl_diagnostics`: count, kinetic energy (f64 sums), bounds of the positions.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct Diagnostics {
    pub count: f64,
    pub kinetic_energy: f64,
    pub pos_max: [f32; 3],
    /// -(min x), -(min y), -(min z): reducible with max like `pos_max`
    pub neg_pos_min: [f32; 3],
    pub pad: [f32; 2],
}

/// `Poly7` of the reference (src/poly7.rs:6-10): `coeffs[i]` multiplies x^i.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct Poly7 {
    pub coeffs: [f32; 8],
}

extern "C" {
    fn pcl_last_error() -> *const c_char;
    fn pcl_create(particles: *const c_void, n: usize, force_grid: *const [f32; 4], n_cells: usize,
                  device: c_int, out: *mut *mut PclCompute) -> c_int;
    fn pcl_destroy(h: *mut PclCompute) -> c_int;
    fn pcl_upload_particles(h: *mut PclCompute, particles: *const c_void, n: usize) -> c_int;
    fn pcl_update_force_grid(h: *mut PclCompute, force_grid: *const [f32; 4], n_cells: usize) -> c_int;
    fn pcl_patch_force_grid(h: *mut PclCompute, first_cell: usize, cells: *const [f32; 4], count: usize) -> c_int;
    fn pcl_update_sim_params(h: *mut PclCompute, params: *const c_void) -> c_int;
    fn pcl_step(h: *mut PclCompute) -> c_int;
    fn pcl_step_dt(h: *mut PclCompute, delta_t: f32) -> c_int;
    fn pcl_read_particles(h: *mut PclCompute, out: *mut c_void, first: usize, count: usize) -> c_int;
    fn pcl_pack_particles_device(h: *mut PclCompute, dev_out: *mut c_void, first: usize, count: usize) -> c_int;
    fn pcl_pack_particles_device_unordered(h: *mut PclCompute, dev_out: *mut c_void) -> c_int;
    fn pcl_step_n(h: *mut PclCompute, n_steps: u64) -> c_int;
    fn pcl_synchronize(h: *mut PclCompute) -> c_int;
    fn pcl_brush_input_default(out: *mut BrushInput);
    fn pcl_brush_mouse_down(h: *mut PclCompute, input: *const BrushInput) -> c_int;
    fn pcl_brush_mouse_moved(h: *mut PclCompute, input: *const BrushInput) -> c_int;
    fn pcl_brush_mouse_up(h: *mut PclCompute) -> c_int;
    fn pcl_read_force_grid(h: *mut PclCompute, out: *mut [f32; 4], first_cell: usize, count: usize) -> c_int;
    fn pcl_patch_force_grid_indexed(h: *mut PclCompute, cell_indices: *const u32, cells: *const [f32; 4], count: usize) -> c_int;
    fn pcl_enable_step_diagnostics(h: *mut PclCompute, enable: c_int) -> c_int;
    fn pcl_diagnostics_read(h: *mut PclCompute, out: *mut Diagnostics) -> c_int;
    fn pcl_device_count() -> c_int;
    fn pcl_poly7_from_points(points: *const [f32; 2], out: *mut Poly7) -> c_int;
    fn pcl_poly7_eval(p: *const Poly7, x: f32) -> f32;
    // several GPUs, one process: RCCL inside the library
    fn pcl_group_create(devices: *const c_int, n_devices: c_int, particles: *const c_void, n: usize,
                        force_grid: *const [f32; 4], n_cells: usize, out: *mut *mut PclGroup) -> c_int;
    fn pcl_group_destroy(g: *mut PclGroup) -> c_int;
    fn pcl_group_size(g: *const PclGroup) -> c_int;
    fn pcl_group_upload_particles(g: *mut PclGroup, particles: *const c_void, n: usize) -> c_int;
    fn pcl_group_update_force_grid(g: *mut PclGroup, force_grid: *const [f32; 4], n_cells: usize) -> c_int;
    fn pcl_group_patch_force_grid(g: *mut PclGroup, first_cell: usize, cells: *const [f32; 4], count: usize) -> c_int;
    fn pcl_group_patch_force_grid_indexed(g: *mut PclGroup, cell_indices: *const u32, cells: *const [f32; 4], count: usize) -> c_int;
    fn pcl_group_update_sim_params(g: *mut PclGroup, params: *const c_void) -> c_int;
    fn pcl_group_enable_step_diagnostics(g: *mut PclGroup, enable: c_int) -> c_int;
    fn pcl_group_step(g: *mut PclGroup) -> c_int;
    fn pcl_group_brush_mouse_down(g: *mut PclGroup, input: *const BrushInput) -> c_int;
    fn pcl_group_brush_mouse_moved(g: *mut PclGroup, input: *const BrushInput) -> c_int;
    fn pcl_group_brush_mouse_up(g: *mut PclGroup) -> c_int;
    fn pcl_group_diagnostics_read(g: *mut PclGroup, out: *mut Diagnostics) -> c_int;
    fn pcl_group_read_particles(g: *mut PclGroup, out: *mut c_void, first: usize, count: usize) -> c_int;
    fn pcl_group_synchronize(g: *mut PclGroup) -> c_int;

    fn pcl_comm_unique_id(out_id: *mut c_void) -> c_int;
    fn pcl_comm_attach(h: *mut PclCompute, unique_id: *const c_void, rank: c_int, world: c_int) -> c_int;
    fn pcl_comm_detach(h: *mut PclCompute) -> c_int;
    fn pcl_comm_count(h: *const PclCompute) -> c_int;
    fn pcl_comm_update_force_grid(h: *mut PclCompute, force_grid: *const [f32; 4], n_cells: usize, root: c_int) -> c_int;
    fn pcl_comm_step(h: *mut PclCompute) -> c_int;
    fn pcl_comm_diagnostics_read(h: *mut PclCompute, out: *mut Diagnostics) -> c_int;
    fn pcl_launch_residency(h: *const PclCompute, intended_blocks_per_cu: *mut u32, granted_blocks_per_cu: *mut u32) -> c_int;
    fn pcl_device_pci_bus_id(h: *const PclCompute, out: *mut c_char, capacity: usize) -> c_int;
    fn pcl_pair_sweep_info(h: *const PclCompute, shared_sweep: *mut u32, cells_occupied: *mut u32) -> c_int;
    fn pcl_debug_state_checksum(h: *mut PclCompute, first_global_index: u64, sum: *mut u64, xor_of_all: *mut u64) -> c_int;
}

impl Poly7 {
    /// `Poly7::from_points` (src/poly7.rs:37-58); `None` where the reference returns `None`.
    pub fn from_points(points: &[[f32; 2]; 8]) -> Option<Poly7> {
        let mut out = Poly7::default();
        match unsafe { pcl_poly7_from_points(points.as_ptr(), &mut out) } {
            0 => Some(out),
            _ => None,
        }
    }

    /// `Poly7::eval` (src/poly7.rs:60-68).
    pub fn eval(&self, x: f32) -> f32 {
        unsafe { pcl_poly7_eval(self, x) }
    }
}

/// Number of HIP devices the library sees.
pub fn device_count() -> usize {
    unsafe { pcl_device_count() }.max(0) as usize
}

/// What `Cursor` and `EditMode` hold per mouse event (src/cursor.rs:5-17,144-151),
/// as the 64-byte record of `pcl_brush_input`.
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct BrushInput {
    pub cursor_pos: [f32; 3],
    pub falloff_dist: f32,
    /// `Cursor::rot.x` (src/cursor.rs:168)
    pub cam_right: [f32; 3],
    pub strength: f32,
    /// `Cursor::rot.z` (src/cursor.rs:167)
    pub cam_dir: [f32; 3],
    /// 0 Centered, 1 Shift, 2 Rotate (src/cursor.rs:124-129)
    pub mode: i32,
    /// 1 Relative, 0 Absolute (src/cursor.rs:131-135)
    pub relative: i32,
    /// 0 Abrupt, 1 Linear, 2 InverseDistance (src/cursor.rs:137-142)
    pub falloff: i32,
    pub reserved: [i32; 2],
}

impl Default for BrushInput {
    /// `Cursor::new` + `EditMode::default`: src/cursor.rs:20-34,206-216.
    fn default() -> Self {
        let mut r = std::mem::MaybeUninit::<BrushInput>::uninit();
        unsafe {
            pcl_brush_input_default(r.as_mut_ptr());
            r.assume_init()
        }
    }
}

fn check(rc: c_int) {
    if rc != 0 {
        // The reference panics on wgpu validation errors; keep that behaviour.
        let msg = unsafe { CStr::from_ptr(pcl_last_error()) }.to_string_lossy().into_owned();
        panic!("particles_hip error {rc}: {msg}");
    }
}

/// `P` is the reference's 48-byte `Particle`, `S` its 912-byte `SimParams`.
pub struct Compute {
    handle: *mut PclCompute,
    pub num_particles: usize,
}

impl Compute {
    /// `Compute::new(device, particles, force_grid)`: src/compute.rs:25, on HIP device 0.
    pub fn new<P: bytemuck::NoUninit>(particles: &[P], force_grid: &[[f32; 4]]) -> Self {
        Self::new_on_device(0, particles, force_grid)
    }

    /// The same on the HIP device with ordinal `device` (the wgpu `&Device` argument's stand-in).
    pub fn new_on_device<P: bytemuck::NoUninit>(device: i32, particles: &[P], force_grid: &[[f32; 4]]) -> Self {
        assert_eq!(std::mem::size_of::<P>(), 48, "Particle must be the reference's 48-byte record");
        let mut handle = std::ptr::null_mut();
        check(unsafe {
            pcl_create(particles.as_ptr() as *const c_void, particles.len(), force_grid.as_ptr(),
                       force_grid.len(), device as c_int, &mut handle)
        });
        Compute { handle, num_particles: particles.len() }
    }

    /// Particles sharded by index range over `devices` (all visible ones when empty), the field
    /// replicated by RCCL broadcast inside the library: same methods, every GPU of the node.
    pub fn new_sharded<P: bytemuck::NoUninit>(devices: &[i32], particles: &[P], force_grid: &[[f32; 4]]) -> ShardedCompute {
        ShardedCompute::new(devices, particles, force_grid)
    }

    /// src/compute.rs:182-196.
    pub fn upload_particles<P: bytemuck::NoUninit>(&mut self, particles: &[P]) {
        assert_eq!(std::mem::size_of::<P>(), 48);
        self.num_particles = particles.len();
        check(unsafe { pcl_upload_particles(self.handle, particles.as_ptr() as *const c_void, particles.len()) });
    }

    /// src/compute.rs:198-204.
    pub fn update_force_grid(&mut self, force_grid: &[[f32; 4]]) {
        check(unsafe { pcl_update_force_grid(self.handle, force_grid.as_ptr(), force_grid.len()) });
    }

    /// Brush edits: only the dirty cells travel (replaces src/main.rs:371-380).
    pub fn patch_force_grid(&mut self, first_cell: usize, cells: &[[f32; 4]]) {
        check(unsafe { pcl_patch_force_grid(self.handle, first_cell, cells.as_ptr(), cells.len()) });
    }

    /// Brush edits as a list of cells (the write loop of src/cursor.rs:84-101).
    pub fn patch_force_grid_indexed(&mut self, cell_indices: &[u32], cells: &[[f32; 4]]) {
        assert_eq!(cell_indices.len(), cells.len());
        check(unsafe { pcl_patch_force_grid_indexed(self.handle, cell_indices.as_ptr(), cells.as_ptr(), cells.len()) });
    }

    /// src/compute.rs:206-219.
    pub fn update_sim_params<S: bytemuck::NoUninit>(&mut self, sim_params: &S) {
        assert_eq!(std::mem::size_of::<S>(), 912, "SimParams must be the reference's 912-byte block");
        check(unsafe { pcl_update_sim_params(self.handle, sim_params as *const S as *const c_void) });
    }

    /// Fuse count / kinetic energy / bounds into every step (no extra pass over the particles).
    pub fn enable_step_diagnostics(&mut self, enable: bool) {
        check(unsafe { pcl_enable_step_diagnostics(self.handle, enable as c_int) });
    }

    /// Count, kinetic energy and bounds of the latest state (synchronises).
    pub fn diagnostics(&mut self) -> Diagnostics {
        let mut d = Diagnostics::default();
        check(unsafe { pcl_diagnostics_read(self.handle, &mut d) });
        d
    }

    /// src/compute.rs:221-227: one update, asynchronous on the HIP stream.
    pub fn compute(&mut self) {
        check(unsafe { pcl_step(self.handle) });
    }

    /// `step(dt)`: set delta_t and run one update.
    pub fn step(&mut self, dt: f32) {
        check(unsafe { pcl_step_dt(self.handle, dt) });
    }

    /// Latest state in caller order as 48-byte records.
    pub fn read_particles<P: bytemuck::Pod>(&mut self, first: usize, count: usize) -> Vec<P> {
        assert_eq!(std::mem::size_of::<P>(), 48);
        let mut out = vec![P::zeroed(); count];
        check(unsafe { pcl_read_particles(self.handle, out.as_mut_ptr() as *mut c_void, first, count) });
        out
    }

    /// Packs the instance stream the renderer draws from (src/renderer.rs:253-257)
    /// into a device buffer shared with the graphics API.
    pub fn pack_particles_device(&mut self, dev_out: *mut c_void, first: usize, count: usize) {
        check(unsafe { pcl_pack_particles_device(self.handle, dev_out, first, count) });
    }

    /// All instance records in memory order (the renderer draws them all, order is irrelevant).
    pub fn pack_particles_device_unordered(&mut self, dev_out: *mut c_void) {
        check(unsafe { pcl_pack_particles_device_unordered(self.handle, dev_out) });
    }

    /// `n` updates without looking at the state in between (headless runs).
    pub fn compute_n(&mut self, n_steps: u64) {
        check(unsafe { pcl_step_n(self.handle, n_steps) });
    }

    pub fn synchronize(&mut self) {
        check(unsafe { pcl_synchronize(self.handle) });
    }

    /// `Cursor::mouse_down` (src/cursor.rs:105-116) on the device field.
    pub fn brush_mouse_down(&mut self, input: &BrushInput) {
        check(unsafe { pcl_brush_mouse_down(self.handle, input) });
    }

    /// `Cursor::mouse_moved` (src/cursor.rs:79-103) on the device field; replaces the
    /// host rewrite and the whole-field upload of src/main.rs:371-380.
    pub fn brush_mouse_moved(&mut self, input: &BrushInput) {
        check(unsafe { pcl_brush_mouse_moved(self.handle, input) });
    }

    /// `Cursor::mouse_up` (src/cursor.rs:118-121).
    pub fn brush_mouse_up(&mut self) {
        check(unsafe { pcl_brush_mouse_up(self.handle) });
    }

    /// The device field, e.g. for drawing the vectors (src/grid.rs:149-170).
    pub fn read_force_grid(&mut self, first_cell: usize, count: usize) -> Vec<[f32; 4]> {
        let mut out = vec![[0.0f32; 4]; count];
        check(unsafe { pcl_read_force_grid(self.handle, out.as_mut_ptr(), first_cell, count) });
        out
    }
}

/// Bytes of the id that rank 0 creates and every rank passes to `Compute::comm_attach`
/// (`PCL_UNIQUE_ID_BYTES`).
pub const UNIQUE_ID_BYTES: usize = 128;

/// One process per GPU: every rank owns one `Compute` holding its index range of the particles and
/// makes the same calls; the library talks RCCL on the handle's own stream.
impl Compute {
    /// The id rank 0 creates; the host carries it to the other ranks by whatever means it has.
    pub fn comm_unique_id() -> [u8; UNIQUE_ID_BYTES] {
        let mut id = [0u8; UNIQUE_ID_BYTES];
        check(unsafe { pcl_comm_unique_id(id.as_mut_ptr() as *mut c_void) });
        id
    }

    /// Joins the communicator of `world` ranks as `rank` (`ncclCommInitRank`).
    pub fn comm_attach(&mut self, unique_id: &[u8; UNIQUE_ID_BYTES], rank: i32, world: i32) {
        check(unsafe { pcl_comm_attach(self.handle, unique_id.as_ptr() as *const c_void, rank as c_int, world as c_int) });
    }

    pub fn comm_detach(&mut self) {
        check(unsafe { pcl_comm_detach(self.handle) });
    }

    /// Ranks the communicator itself counts (`ncclCommCount`); 0 without a communicator.
    pub fn comm_count(&self) -> i32 {
        unsafe { pcl_comm_count(self.handle) }
    }

    /// `update_force_grid` on every rank: `root`'s field is broadcast into every replica (the
    /// other ranks pass a slice of the same length, whose contents are ignored).
    pub fn comm_update_force_grid(&mut self, force_grid: &[[f32; 4]], root: i32) {
        check(unsafe { pcl_comm_update_force_grid(self.handle, force_grid.as_ptr(), force_grid.len(), root as c_int) });
    }

    /// `compute()` of this rank's range, with the all-gather of the pair term and the all-reduce of
    /// the step diagnostics where those are active.  All ranks upload new particle sets and change
    /// these two settings between the same two steps; the first step after such a change (and after
    /// attaching) agrees on them with one all-reduce that is read back on the host, and fails on every
    /// rank when the ranks differ.
    pub fn comm_compute(&mut self) {
        check(unsafe { pcl_comm_step(self.handle) });
    }

    /// Diagnostics over all ranks (synchronises this rank's stream).
    pub fn comm_diagnostics(&mut self) -> Diagnostics {
        let mut d = Diagnostics::default();
        check(unsafe { pcl_comm_diagnostics_read(self.handle, &mut d) });
        d
    }

    /// Resident blocks per CU the streaming step asked for and what the runtime's occupancy
    /// calculation grants (both 0 before the first capped launch).
    pub fn launch_residency(&self) -> (u32, u32) {
        let (mut intended, mut granted) = (0u32, 0u32);
        check(unsafe { pcl_launch_residency(self.handle, &mut intended, &mut granted) });
        (intended, granted)
    }

    /// Which of the two (bit-identical) sweeps of the cell-binned pair term the last step ran -- `true`:
    /// candidates shared by a wave -- and the count of occupied cut-off cells it was chosen from.
    pub fn pair_sweep_info(&self) -> (bool, u32) {
        let (mut shared, mut cells) = (0u32, 0u32);
        check(unsafe { pcl_pair_sweep_info(self.handle, &mut shared, &mut cells) });
        (shared != 0, cells)
    }

    /// Order-independent checksum (sum, xor) of the whole state; the words of index ranges, each taken
    /// with its first global index, combine to those of one handle holding the whole set.
    pub fn state_checksum(&mut self, first_global_index: u64) -> (u64, u64) {
        let (mut sum, mut xor) = (0u64, 0u64);
        check(unsafe { pcl_debug_state_checksum(self.handle, first_global_index, &mut sum, &mut xor) });
        (sum, xor)
    }

    /// PCI bus id of the handle's GPU ("0000:c1:00.0"), to tell devices apart across processes.
    pub fn pci_bus_id(&self) -> String {
        let mut buf = [0 as c_char; 32];
        check(unsafe { pcl_device_pci_bus_id(self.handle, buf.as_mut_ptr(), buf.len()) });
        unsafe { CStr::from_ptr(buf.as_ptr()) }.to_string_lossy().into_owned()
    }
}

impl Drop for Compute {
    fn drop(&mut self) {
        unsafe { pcl_destroy(self.handle) };
    }
}

/// `Compute` over several GPUs of one node driven by this process (`pcl_group_*`): particles
/// shard by contiguous index range, the field replicas are kept equal by RCCL broadcasts, and
/// with step diagnostics on every step all-reduces the 48-byte record.
pub struct ShardedCompute {
    group: *mut PclGroup,
    pub num_particles: usize,
}

impl ShardedCompute {
    pub fn new<P: bytemuck::NoUninit>(devices: &[i32], particles: &[P], force_grid: &[[f32; 4]]) -> Self {
        assert_eq!(std::mem::size_of::<P>(), 48, "Particle must be the reference's 48-byte record");
        let all: Vec<c_int> = if devices.is_empty() {
            (0..device_count().max(1) as c_int).collect()
        } else {
            devices.iter().map(|&d| d as c_int).collect()
        };
        let mut group = std::ptr::null_mut();
        check(unsafe {
            pcl_group_create(all.as_ptr(), all.len() as c_int, particles.as_ptr() as *const c_void, particles.len(),
                             force_grid.as_ptr(), force_grid.len(), &mut group)
        });
        ShardedCompute { group, num_particles: particles.len() }
    }

    pub fn num_gpus(&self) -> usize {
        unsafe { pcl_group_size(self.group) }.max(0) as usize
    }

    pub fn upload_particles<P: bytemuck::NoUninit>(&mut self, particles: &[P]) {
        assert_eq!(std::mem::size_of::<P>(), 48);
        self.num_particles = particles.len();
        check(unsafe { pcl_group_upload_particles(self.group, particles.as_ptr() as *const c_void, particles.len()) });
    }

    pub fn update_force_grid(&mut self, force_grid: &[[f32; 4]]) {
        check(unsafe { pcl_group_update_force_grid(self.group, force_grid.as_ptr(), force_grid.len()) });
    }

    pub fn patch_force_grid(&mut self, first_cell: usize, cells: &[[f32; 4]]) {
        check(unsafe { pcl_group_patch_force_grid(self.group, first_cell, cells.as_ptr(), cells.len()) });
    }

    pub fn patch_force_grid_indexed(&mut self, cell_indices: &[u32], cells: &[[f32; 4]]) {
        assert_eq!(cell_indices.len(), cells.len());
        check(unsafe { pcl_group_patch_force_grid_indexed(self.group, cell_indices.as_ptr(), cells.as_ptr(), cells.len()) });
    }

    pub fn update_sim_params<S: bytemuck::NoUninit>(&mut self, sim_params: &S) {
        assert_eq!(std::mem::size_of::<S>(), 912);
        check(unsafe { pcl_group_update_sim_params(self.group, sim_params as *const S as *const c_void) });
    }

    pub fn enable_step_diagnostics(&mut self, enable: bool) {
        check(unsafe { pcl_group_enable_step_diagnostics(self.group, enable as c_int) });
    }

    /// One update of every range (src/compute.rs:221-227 on every GPU).
    pub fn compute(&mut self) {
        check(unsafe { pcl_group_step(self.group) });
    }

    pub fn brush_mouse_down(&mut self, input: &BrushInput) {
        check(unsafe { pcl_group_brush_mouse_down(self.group, input) });
    }

    pub fn brush_mouse_moved(&mut self, input: &BrushInput) {
        check(unsafe { pcl_group_brush_mouse_moved(self.group, input) });
    }

    pub fn brush_mouse_up(&mut self) {
        check(unsafe { pcl_group_brush_mouse_up(self.group) });
    }

    /// Diagnostics of the whole set (all-reduced over the GPUs).
    pub fn diagnostics(&mut self) -> Diagnostics {
        let mut d = Diagnostics::default();
        check(unsafe { pcl_group_diagnostics_read(self.group, &mut d) });
        d
    }

    pub fn read_particles<P: bytemuck::Pod>(&mut self, first: usize, count: usize) -> Vec<P> {
        assert_eq!(std::mem::size_of::<P>(), 48);
        let mut out = vec![P::zeroed(); count];
        check(unsafe { pcl_group_read_particles(self.group, out.as_mut_ptr() as *mut c_void, first, count) });
        out
    }

    pub fn synchronize(&mut self) {
        check(unsafe { pcl_group_synchronize(self.group) });
    }
}

impl Drop for ShardedCompute {
    fn drop(&mut self) {
        unsafe { pcl_group_destroy(self.group) };
    }
}
