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
    /// `Compute::new(device, particles, force_grid)`: src/compute.rs:25.
    pub fn new<P: bytemuck::NoUninit>(particles: &[P], force_grid: &[[f32; 4]]) -> Self {
        assert_eq!(std::mem::size_of::<P>(), 48, "Particle must be the reference's 48-byte record");
        let mut handle = std::ptr::null_mut();
        check(unsafe {
            pcl_create(particles.as_ptr() as *const c_void, particles.len(), force_grid.as_ptr(),
                       force_grid.len(), 0, &mut handle)
        });
        Compute { handle, num_particles: particles.len() }
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

    /// src/compute.rs:206-219.
    pub fn update_sim_params<S: bytemuck::NoUninit>(&mut self, sim_params: &S) {
        assert_eq!(std::mem::size_of::<S>(), 912, "SimParams must be the reference's 912-byte block");
        check(unsafe { pcl_update_sim_params(self.handle, sim_params as *const S as *const c_void) });
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

impl Drop for Compute {
    fn drop(&mut self) {
        unsafe { pcl_destroy(self.handle) };
    }
}
