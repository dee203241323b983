//! sfx-sys: the thin `extern "C"` FFI crate BASELINE.json:5 asks for, mirroring include/sfx.h.
//! UNCOMPILED: no rustc/cargo in the build image and no upstream source to test against; the
//! C harnesses (starframe_amd/cpp, ctypes) exercise the identical ABI.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int};

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct sfx_config {
    pub struct_size: u32,
    pub device: i32,
    pub substeps: u32,
    pub gravity: [f32; 2],
    pub aabb_margin: f32,
    pub vel_pad_max: f32,
    pub cell_size: f32,
    pub grid_log2: [u32; 2],
    pub max_pairs: u32,
    pub color_rounds: u32,
    pub tile_cells: u32,
    pub tile_cells_y: u32,
    pub restitution_threshold: f32,
    pub max_depenetration_velocity: f32,
    pub speculative_margin: f32,
    pub linear_damping: f32,
    pub angular_damping: f32,
    pub flags: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct sfx_collider {
    pub type_: u32,
    pub material: u32,
    pub category: u32,
    pub mask: u32,
    pub group: u32,
    pub p: [f32; 3],
    pub poly_offset: u32,
    pub poly_count: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct sfx_material {
    pub restitution: f32,
    pub static_friction: f32,
    pub dynamic_friction: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct sfx_distance_constraint {
    pub body_a: u32,
    pub body_b: u32,
    pub anchor_a: [f32; 2],
    pub anchor_b: [f32; 2],
    pub rest_length: f32,
    pub compliance: f32,
    pub limit: u32,    // SFX_LIMIT_*: 0 both ways, 1 rope (only pulls), 2 strut (only pushes)
    pub damping: f32,  // 1/s, two-sided constraints only
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct sfx_angle_constraint {
    pub body_a: u32,
    pub body_b: u32,
    pub min_angle: f32, // -pi < min <= max < pi, relative angle b - a
    pub max_angle: f32,
    pub compliance: f32,
    pub damping: f32,
}

#[repr(C)]
pub struct sfx_world {
    _private: [u8; 0],
}

pub const SFX_ABI_VERSION: u32 = 7;
pub const SFX_OK: c_int = 0;
pub const SFX_WORLD_BODY: u32 = 0xFFFF_FFFF;

pub const SFX_NO_BODY: u32 = 0xFFFF_FFFF;

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct sfx_ray_hit {
    pub body: u32,
    pub t: f32,
    pub normal: [f32; 2],
}

extern "C" {
    pub fn sfx_default_config(cfg: *mut sfx_config);
    pub fn sfx_create(cfg: *const sfx_config, out: *mut *mut sfx_world) -> c_int;
    pub fn sfx_destroy(w: *mut sfx_world);
    pub fn sfx_upload_bodies(w: *mut sfx_world, n: u32, pos: *const f32, rot: *const f32, vel: *const f32,
                             angvel: *const f32, inv_mass: *const f32, inv_inertia: *const f32) -> c_int;
    pub fn sfx_upload_colliders(w: *mut sfx_world, n: u32, c: *const sfx_collider, poly_verts: *const f32,
                                n_poly_verts: u32) -> c_int;
    pub fn sfx_upload_materials(w: *mut sfx_world, n: u32, m: *const sfx_material) -> c_int;
    pub fn sfx_upload_constraints(w: *mut sfx_world, m: u32, c: *const sfx_distance_constraint) -> c_int;
    pub fn sfx_readback_begin(w: *mut sfx_world) -> c_int;
    pub fn sfx_readback_wait(w: *mut sfx_world, poses: *mut *const f32, n: *mut u32) -> c_int;
    pub fn sfx_upload_angle_constraints(w: *mut sfx_world, m: u32, c: *const sfx_angle_constraint) -> c_int;
    pub fn sfx_step(w: *mut sfx_world, frame_dt: f32) -> c_int;
    pub fn sfx_sync(w: *mut sfx_world) -> c_int;
    pub fn sfx_abi_version() -> u32;
    // the tick in pieces + slab halos (one world per GPU, boundary bodies swapped every substep)
    pub fn sfx_tick_begin(w: *mut sfx_world, frame_dt: f32) -> c_int;
    pub fn sfx_tick_substep(w: *mut sfx_world) -> c_int;
    pub fn sfx_set_stream(w: *mut sfx_world, hip_stream: *mut std::os::raw::c_void) -> c_int;
    pub fn sfx_halo_set(w: *mut sfx_world, send_idx: *const u32, n_send: u32, recv_idx: *const u32, n_recv: u32) -> c_int;
    pub fn sfx_halo_pack(w: *mut sfx_world) -> c_int;
    pub fn sfx_halo_unpack(w: *mut sfx_world) -> c_int;
    pub fn sfx_halo_buffers(w: *mut sfx_world, send_dev: *mut *mut f32, recv_dev: *mut *mut f32) -> c_int;
    pub fn sfx_halo_download(w: *mut sfx_world, send_host: *mut f32) -> c_int;
    pub fn sfx_halo_upload(w: *mut sfx_world, recv_host: *const f32) -> c_int;
    pub fn sfx_halo_copy(dst: *mut sfx_world, dst_first: u32, src: *mut sfx_world, src_first: u32, count: u32) -> c_int;
    pub fn sfx_download_poses(w: *mut sfx_world, pos: *mut f32, rot: *mut f32) -> c_int;
    pub fn sfx_download_velocities(w: *mut sfx_world, vel: *mut f32, angvel: *mut f32) -> c_int;
    pub fn sfx_last_error(w: *mut sfx_world) -> *const c_char;
    // spatial queries at the current poses (include/sfx.h "spatial queries")
    pub fn sfx_query_points(w: *mut sfx_world, n: u32, points: *const f32, mask: u32, body: *mut u32, count: *mut u32) -> c_int;
    pub fn sfx_query_aabbs(w: *mut sfx_world, n: u32, boxes: *const f32, mask: u32, hits: *mut u64, cap: usize,
                           n_hits: *mut usize) -> c_int;
    pub fn sfx_query_rays(w: *mut sfx_world, n: u32, rays: *const f32, max_t: *const f32, mask: u32,
                          hits: *mut sfx_ray_hit) -> c_int;
}

/// Owning handle; `Drop` calls `sfx_destroy`.  A starframe-side `Physics::tick(&mut self, dt)` marshals
/// its body / collider / material components into the SoA slices once (or when they change) and
/// then calls `step` once per frame.
pub struct World(*mut sfx_world);

impl World {
    pub fn new(cfg: &sfx_config) -> Result<Self, String> {
        let mut w = std::ptr::null_mut();
        let rc = unsafe { sfx_create(cfg, &mut w) };
        if rc == SFX_OK { Ok(World(w)) } else { Err(last_error(std::ptr::null_mut())) }
    }
    pub fn step(&mut self, frame_dt: f32) -> Result<(), String> {
        let rc = unsafe { sfx_step(self.0, frame_dt) };
        if rc == SFX_OK { Ok(()) } else { Err(last_error(self.0)) }
    }
    pub fn raw(&mut self) -> *mut sfx_world { self.0 }
}

impl Drop for World {
    fn drop(&mut self) { unsafe { sfx_destroy(self.0) } }
}

fn last_error(w: *mut sfx_world) -> String {
    unsafe { std::ffi::CStr::from_ptr(sfx_last_error(w)).to_string_lossy().into_owned() }
}
