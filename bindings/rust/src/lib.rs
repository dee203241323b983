//! sfx-sys: the thin `extern "C"` FFI crate BASELINE.json:5 asks for, mirroring include/sfx.h and
//! include/sfx_rccl.h declaration by declaration (ABI 10).
//! UNCOMPILED: no rustc/cargo in the build image and no upstream source to test against; the C and C++
//! harnesses (tools/sfx_c_tick.c, starframe_amd/cpp, tools/sfx_slab_run.cpp) and ctypes exercise the identical
//! ABI, and tests/test_abi.py::test_rust_mirror_declares_every_symbol_and_size checks this file's text against
//! the headers: every exported symbol is declared here and every struct size below is the header's.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

pub const SFX_ABI_VERSION: u32 = 12;

// sfx_status
pub const SFX_OK: c_int = 0;
pub const SFX_ERR_INVALID: c_int = -1;
pub const SFX_ERR_UNSUPPORTED: c_int = -2;
pub const SFX_ERR_DEVICE: c_int = -3;
pub const SFX_ERR_CAPACITY: c_int = -4;
pub const SFX_ERR_NOMEM: c_int = -5;

// sfx_shape_type
pub const SFX_SHAPE_NONE: u32 = 0;
pub const SFX_SHAPE_CIRCLE: u32 = 1;
pub const SFX_SHAPE_RECT: u32 = 2;
pub const SFX_SHAPE_CAPSULE: u32 = 3;
pub const SFX_SHAPE_POLY: u32 = 4;

pub const SFX_MAX_POLY_VERTS: u32 = 8;
pub const SFX_WORLD_BODY: u32 = 0xFFFF_FFFF;
pub const SFX_NO_BODY: u32 = 0xFFFF_FFFF;
pub const SFX_FALLBACK_TILE: u32 = 0xFFFF_FFFF;
pub const SFX_MAX_COLORS: usize = 64;
pub const SFX_MAX_TAN_CORRECTION: f32 = 0.25;
pub const SFX_TILE_MAX_PAIRS: u32 = 2048;
pub const SFX_TILE_MAX_BODIES: u32 = 1024;
pub const SFX_TILE_WHOLE_GRID: u32 = 0xFFFF;
pub const SFX_ONE_TILE_COLLIDERS: u32 = 512;
pub const SFX_ONE_TILE_LARGE: u32 = 64;
pub const SFX_CFG_MULTI_LAUNCH: u32 = 1;
pub const SFX_CFG_NO_FRICTION: u32 = 2;
pub const SFX_CFG_ONE_LEVEL: u32 = 4;
pub const SFX_CFG_ONE_WORKGROUP: u32 = 8;
pub const SFX_LIMIT_NONE: u32 = 0;
pub const SFX_LIMIT_ROPE: u32 = 1;
pub const SFX_LIMIT_STRUT: u32 = 2;

// sfx_kernel_class
pub const SFX_K_AABB_COUNT: usize = 0;
pub const SFX_K_SCAN: usize = 1;
pub const SFX_K_SCATTER: usize = 2;
pub const SFX_K_PAIR_COUNT: usize = 3;
pub const SFX_K_PAIR_FILL: usize = 4;
pub const SFX_K_TILE_LIST: usize = 5;
pub const SFX_K_COLOR: usize = 6;
pub const SFX_K_INTEGRATE: usize = 7;
pub const SFX_K_SOLVE_JOINTS: usize = 8;
pub const SFX_K_SOLVE_POS: usize = 9;
pub const SFX_K_DERIVE_VEL: usize = 10;
pub const SFX_K_SOLVE_VEL: usize = 11;
pub const SFX_K_STATS: usize = 12;
pub const SFX_K_HALO: usize = 13;
pub const SFX_K_QUERY: usize = 14;
pub const SFX_K_NARROW: usize = 15;
pub const SFX_K_TICK: usize = 16;
pub const SFX_K_CLASSES: usize = 17;

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
    pub tile_cells: u32,   // 2, 4 or 8 grid cells, or SFX_TILE_WHOLE_GRID; 0 = automatic (one tile up to 512 colliders, else 2)
    pub tile_cells_y: u32, // 0 = automatic (strips for wide, shallow scenes)
    pub restitution_threshold: f32,
    pub max_depenetration_velocity: f32, // 0 = unlimited
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
    pub local_pos: [f32; 2], // the shape's origin in the body frame
    pub local_angle: f32,    // and its rotation there (radians)
    pub body: u32,           // colliders beyond the body count: the body this one belongs to
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
    pub limit: u32,   // SFX_LIMIT_*
    pub damping: f32, // 1/s, two-sided constraints only
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
#[derive(Clone, Copy, Debug, Default)]
pub struct sfx_contact {
    pub body_a: u32,
    pub body_b: u32,
    pub count: u32,
    pub normal: [f32; 2],
    pub depth: [f32; 2],
    pub point_a: [[f32; 2]; 2],
    pub point_b: [[f32; 2]; 2],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct sfx_frame_stats {
    pub n_bodies: u32,
    pub n_small: u32,
    pub n_large: u32,
    pub n_pairs: u32,
    pub n_colors: u32,
    pub n_leftover: u32,
    pub n_manifolds: u32,
    pub n_manifolds2: u32,
    pub n_contacts: u32,
    pub n_joint_colors: u32,
    pub overflow: u32,
    pub launches: u32,
    pub n_tiles: u32,
    pub n_fallback: u32,
    pub max_tile_pairs: u32,
    pub max_tile_bodies: u32,
    pub n_solver_pairs: u32,
    pub max_depth: f32,
    pub kinetic_energy: f32,
    pub color_count: [u32; SFX_MAX_COLORS],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct sfx_ray_hit {
    pub body: u32,
    pub t: f32,
    pub normal: [f32; 2],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct sfx_kernel_time {
    pub total_ms: f64,
    pub launches: u64,
    pub items: u64,
}

#[repr(C)]
pub struct sfx_world {
    _private: [u8; 0],
}

#[repr(C)]
pub struct sfx_rccl_comm {
    _private: [u8; 0],
}

pub const SFX_RCCL_ID_BYTES: usize = 128;

// The sizes include/sfx.h asserts (SFX_STATIC_ASSERT at its end): a mismatch fails the build of this crate.
const _: () = assert!(std::mem::size_of::<sfx_config>() == 80);
const _: () = assert!(std::mem::size_of::<sfx_collider>() == 56);
const _: () = assert!(std::mem::size_of::<sfx_material>() == 12);
const _: () = assert!(std::mem::size_of::<sfx_distance_constraint>() == 40);
const _: () = assert!(std::mem::size_of::<sfx_angle_constraint>() == 24);
const _: () = assert!(std::mem::size_of::<sfx_contact>() == 60);
const _: () = assert!(std::mem::size_of::<sfx_frame_stats>() == 76 + 4 * SFX_MAX_COLORS);
const _: () = assert!(std::mem::size_of::<sfx_ray_hit>() == 16);
const _: () = assert!(std::mem::size_of::<sfx_kernel_time>() == 24);

#[link(name = "sfx")]
extern "C" {
    // lifetime
    pub fn sfx_create(cfg: *const sfx_config, out: *mut *mut sfx_world) -> c_int;
    pub fn sfx_destroy(w: *mut sfx_world);
    pub fn sfx_abi_version() -> u32;
    pub fn sfx_build_id() -> *const c_char;
    pub fn sfx_default_config(cfg: *mut sfx_config);
    // scene upload
    pub fn sfx_upload_bodies(w: *mut sfx_world, n: u32, pos: *const f32, rot: *const f32, vel: *const f32,
                             angvel: *const f32, inv_mass: *const f32, inv_inertia: *const f32) -> c_int;
    pub fn sfx_upload_colliders(w: *mut sfx_world, n: u32, colliders: *const sfx_collider, poly_verts: *const f32,
                                n_poly_verts: u32) -> c_int;
    pub fn sfx_upload_materials(w: *mut sfx_world, n: u32, materials: *const sfx_material) -> c_int;
    pub fn sfx_upload_constraints(w: *mut sfx_world, m: u32, c: *const sfx_distance_constraint) -> c_int;
    pub fn sfx_upload_angle_constraints(w: *mut sfx_world, m: u32, c: *const sfx_angle_constraint) -> c_int;
    // the tick
    pub fn sfx_step(w: *mut sfx_world, frame_dt: f32) -> c_int;
    pub fn sfx_sync(w: *mut sfx_world) -> c_int;
    // the tick in pieces + slab halos
    pub fn sfx_tick_begin(w: *mut sfx_world, frame_dt: f32) -> c_int;
    pub fn sfx_tick_substep(w: *mut sfx_world) -> c_int;
    pub fn sfx_set_stream(w: *mut sfx_world, hip_stream: *mut c_void) -> c_int;
    pub fn sfx_get_stream(w: *mut sfx_world, hip_stream: *mut *mut c_void) -> c_int;
    pub fn sfx_halo_counts(w: *mut sfx_world, n_send: *mut u32, n_recv: *mut u32) -> c_int;
    pub fn sfx_halo_set(w: *mut sfx_world, send_idx: *const u32, n_send: u32, recv_idx: *const u32, n_recv: u32) -> c_int;
    pub fn sfx_halo_guard(w: *mut sfx_world, max_move: f32) -> c_int;
    pub fn sfx_halo_pack(w: *mut sfx_world) -> c_int;
    pub fn sfx_halo_unpack(w: *mut sfx_world) -> c_int;
    pub fn sfx_halo_buffers(w: *mut sfx_world, send_dev: *mut *mut f32, recv_dev: *mut *mut f32) -> c_int;
    pub fn sfx_halo_download(w: *mut sfx_world, send_host: *mut f32) -> c_int;
    pub fn sfx_halo_upload(w: *mut sfx_world, recv_host: *const f32) -> c_int;
    pub fn sfx_halo_copy(dst: *mut sfx_world, dst_first: u32, src: *mut sfx_world, src_first: u32, count: u32) -> c_int;
    pub fn sfx_halo_overlap(w: *mut sfx_world, on: c_int) -> c_int;
    pub fn sfx_halo_stream(w: *mut sfx_world, hip_stream: *mut *mut c_void) -> c_int;
    // state read-back
    pub fn sfx_download_poses(w: *mut sfx_world, pos: *mut f32, rot: *mut f32) -> c_int;
    pub fn sfx_download_velocities(w: *mut sfx_world, vel: *mut f32, angvel: *mut f32) -> c_int;
    pub fn sfx_readback_begin(w: *mut sfx_world) -> c_int;
    pub fn sfx_readback_wait(w: *mut sfx_world, poses: *mut *const f32, n: *mut u32) -> c_int;
    // spatial queries at the current poses
    pub fn sfx_query_points(w: *mut sfx_world, n: u32, points: *const f32, mask: u32, body: *mut u32, count: *mut u32) -> c_int;
    pub fn sfx_query_aabbs(w: *mut sfx_world, n: u32, boxes: *const f32, mask: u32, hits: *mut u64, cap: usize,
                           n_hits: *mut usize) -> c_int;
    pub fn sfx_query_rays(w: *mut sfx_world, n: u32, rays: *const f32, max_t: *const f32, mask: u32,
                          hits: *mut sfx_ray_hit) -> c_int;
    /// Device pointers in and out; allocates nothing, does not synchronise (include/sfx.h).
    pub fn sfx_query_points_static(w: *mut sfx_world, n: u32, points_dev: *const f32, mask: u32, body_dev: *mut u32,
                                   count_dev: *mut u32) -> c_int;
    /// Device pointers in and out, enqueued on the world's stream, no synchronisation (include/sfx.h).
    pub fn sfx_query_aabbs_device(w: *mut sfx_world, n: u32, boxes_dev: *const f32, mask: u32, hits_dev: *mut u64, cap: u32,
                                  n_hits_dev: *mut u32) -> c_int;
    pub fn sfx_query_rays_device(w: *mut sfx_world, n: u32, rays_dev: *const f32, max_t_dev: *const f32, mask: u32,
                                 hits_dev: *mut sfx_ray_hit) -> c_int;
    // parity hooks
    pub fn sfx_debug_aabbs(w: *mut sfx_world, out: *mut f32) -> c_int;
    pub fn sfx_debug_pairs(w: *mut sfx_world, keys: *mut u64, cap: usize, n: *mut usize) -> c_int;
    pub fn sfx_debug_pair_colors(w: *mut sfx_world, keys: *mut u64, colors: *mut u8, tiles: *mut u32, cap: usize,
                                 n: *mut usize) -> c_int;
    pub fn sfx_debug_broadphase(w: *mut sfx_world, frame_dt: f32) -> c_int;
    pub fn sfx_debug_contacts(w: *mut sfx_world, out: *mut sfx_contact, cap: usize, n: *mut usize) -> c_int;
    pub fn sfx_debug_joint_order(w: *mut sfx_world, perm: *mut u32, colors: *mut u8, cap: usize, n: *mut usize) -> c_int;
    pub fn sfx_debug_near_batches(w: *mut sfx_world, n_near: *mut u32, n_batches: *mut u32) -> c_int;
    // reporting
    pub fn sfx_stats(w: *mut sfx_world, out: *mut sfx_frame_stats) -> c_int;
    pub fn sfx_set_profiling(w: *mut sfx_world, on: c_int) -> c_int;
    pub fn sfx_kernel_times(w: *mut sfx_world, out: *mut sfx_kernel_time, reset: c_int) -> c_int;
    pub fn sfx_kernel_class_name(cls: c_int) -> *const c_char;
    pub fn sfx_timer_begin(w: *mut sfx_world) -> c_int;
    pub fn sfx_timer_end(w: *mut sfx_world, ms: *mut f64) -> c_int;
    pub fn sfx_last_error(w: *mut sfx_world) -> *const c_char;
}

// include/sfx_rccl.h: the slab halo exchange over plain rccl.h, one process per GPU
#[link(name = "sfx_rccl")]
extern "C" {
    pub fn sfx_rccl_unique_id(id: *mut u8) -> c_int;
    pub fn sfx_rccl_comm_create(id: *const u8, rank: c_int, nranks: c_int, device: c_int, out: *mut *mut sfx_rccl_comm) -> c_int;
    pub fn sfx_rccl_comm_destroy(c: *mut sfx_rccl_comm);
    pub fn sfx_rccl_comm_abort(c: *mut sfx_rccl_comm);
    pub fn sfx_rccl_halo_exchange(c: *mut sfx_rccl_comm, w: *mut sfx_world, left: c_int, right: c_int, n_send_left: u32,
                                  n_send_right: u32, n_recv_left: u32, n_recv_right: u32) -> c_int;
    pub fn sfx_rccl_sendrecv(c: *mut sfx_rccl_comm, hip_stream: *mut c_void, send_dev: *const f32, n_send: usize, to: c_int,
                             recv_dev: *mut f32, n_recv: usize, from: c_int) -> c_int;
    pub fn sfx_rccl_allreduce_stats(c: *mut sfx_rccl_comm, sums: *mut f64, n_sum: u32, maxs: *mut f64, n_max: u32) -> c_int;
    pub fn sfx_rccl_last_error() -> *const c_char;
}

/// What a call returned when it was not SFX_OK: the status and the library's message.
#[derive(Debug, Clone)]
pub struct Error {
    pub status: c_int,
    pub message: String,
}

impl std::fmt::Display for Error {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "sfx error {}: {}", self.status, self.message)
    }
}
impl std::error::Error for Error {}

pub type Result<T> = std::result::Result<T, Error>;

fn check(w: *mut sfx_world, rc: c_int) -> Result<()> {
    if rc == SFX_OK {
        return Ok(());
    }
    let message = unsafe { std::ffi::CStr::from_ptr(sfx_last_error(w)).to_string_lossy().into_owned() };
    Err(Error { status: rc, message })
}

/// The body state a starframe-side `Physics` marshals out of its components: structure of arrays, one
/// entry per body, `pos` / `rot` / `vel` as (x, y) pairs.  `rot` is (cos, sin) of the body angle.
#[derive(Default, Clone)]
pub struct Bodies {
    pub pos: Vec<[f32; 2]>,
    pub rot: Vec<[f32; 2]>,
    pub vel: Vec<[f32; 2]>,
    pub angvel: Vec<f32>,
    pub inv_mass: Vec<f32>,
    pub inv_inertia: Vec<f32>,
}

/// Owning handle; `Drop` calls `sfx_destroy`.  A world is `!Sync` by construction (raw pointer) — the
/// reference's tick takes `&mut self`, and so does every call here that enqueues work.
pub struct World(*mut sfx_world);

impl World {
    pub fn default_config() -> sfx_config {
        let mut cfg = sfx_config::default();
        unsafe { sfx_default_config(&mut cfg) };
        cfg
    }

    pub fn new(cfg: &sfx_config) -> Result<Self> {
        if unsafe { sfx_abi_version() } != SFX_ABI_VERSION {
            return Err(Error { status: SFX_ERR_INVALID, message: "libsfx has another ABI version than this crate".into() });
        }
        let mut w = std::ptr::null_mut();
        let rc = unsafe { sfx_create(cfg, &mut w) };
        check(std::ptr::null_mut(), rc)?;
        Ok(World(w))
    }

    /// Bodies, then colliders (one per body), then materials; constraints are optional.
    pub fn upload_bodies(&mut self, b: &Bodies) -> Result<()> {
        let n = b.pos.len();
        assert!(b.rot.len() == n && b.vel.len() == n && b.angvel.len() == n && b.inv_mass.len() == n && b.inv_inertia.len() == n);
        let rc = unsafe {
            sfx_upload_bodies(self.0, n as u32, b.pos.as_ptr() as *const f32, b.rot.as_ptr() as *const f32,
                              b.vel.as_ptr() as *const f32, b.angvel.as_ptr(), b.inv_mass.as_ptr(), b.inv_inertia.as_ptr())
        };
        check(self.0, rc)
    }

    pub fn upload_colliders(&mut self, colliders: &[sfx_collider], poly_verts: &[[f32; 2]]) -> Result<()> {
        let rc = unsafe {
            sfx_upload_colliders(self.0, colliders.len() as u32, colliders.as_ptr(), poly_verts.as_ptr() as *const f32,
                                 poly_verts.len() as u32)
        };
        check(self.0, rc)
    }

    pub fn upload_materials(&mut self, materials: &[sfx_material]) -> Result<()> {
        check(self.0, unsafe { sfx_upload_materials(self.0, materials.len() as u32, materials.as_ptr()) })
    }

    pub fn upload_constraints(&mut self, c: &[sfx_distance_constraint]) -> Result<()> {
        check(self.0, unsafe { sfx_upload_constraints(self.0, c.len() as u32, c.as_ptr()) })
    }

    pub fn upload_angle_constraints(&mut self, c: &[sfx_angle_constraint]) -> Result<()> {
        check(self.0, unsafe { sfx_upload_angle_constraints(self.0, c.len() as u32, c.as_ptr()) })
    }

    /// The per-frame tick: asynchronous.  `Err` with `SFX_ERR_CAPACITY` means an earlier tick overflowed a
    /// device buffer and `sync` has not reported it yet (include/sfx.h, sfx_sync).
    pub fn step(&mut self, frame_dt: f32) -> Result<()> {
        check(self.0, unsafe { sfx_step(self.0, frame_dt) })
    }

    /// Waits for everything enqueued; reports — once — a capacity overflow of any tick since the last sync.
    pub fn sync(&mut self) -> Result<()> {
        check(self.0, unsafe { sfx_sync(self.0) })
    }

    /// Poses into caller-owned slices of `n` bodies each (synchronises).
    pub fn download_poses(&mut self, pos: &mut [[f32; 2]], rot: &mut [[f32; 2]]) -> Result<()> {
        assert!(pos.len() == rot.len());
        check(self.0, unsafe { sfx_download_poses(self.0, pos.as_mut_ptr() as *mut f32, rot.as_mut_ptr() as *mut f32) })
    }

    pub fn download_velocities(&mut self, vel: &mut [[f32; 2]], angvel: &mut [f32]) -> Result<()> {
        assert!(vel.len() == angvel.len());
        check(self.0, unsafe { sfx_download_velocities(self.0, vel.as_mut_ptr() as *mut f32, angvel.as_mut_ptr()) })
    }

    /// Render loop: `readback_begin(); step(dt); readback_wait()` — the poses of the frame before travel while
    /// the next tick runs.  The slice is library-owned pinned memory, valid until the second `readback_begin`
    /// after the one it belongs to.
    pub fn readback_begin(&mut self) -> Result<()> {
        check(self.0, unsafe { sfx_readback_begin(self.0) })
    }

    pub fn readback_wait(&mut self) -> Result<&[[f32; 4]]> {
        let mut p: *const f32 = std::ptr::null();
        let mut n: u32 = 0;
        check(self.0, unsafe { sfx_readback_wait(self.0, &mut p, &mut n) })?;
        Ok(unsafe { std::slice::from_raw_parts(p as *const [f32; 4], n as usize) })
    }

    /// Body under a point (lowest id when several), at the current poses.
    pub fn query_point(&mut self, point: [f32; 2], mask: u32) -> Result<Option<u32>> {
        let mut body = SFX_NO_BODY;
        check(self.0, unsafe { sfx_query_points(self.0, 1, point.as_ptr(), mask, &mut body, std::ptr::null_mut()) })?;
        Ok(if body == SFX_NO_BODY { None } else { Some(body) })
    }

    /// Closest hit of `origin + t * dir`, `0 <= t <= max_t`.
    pub fn raycast(&mut self, origin: [f32; 2], dir: [f32; 2], max_t: f32, mask: u32) -> Result<Option<sfx_ray_hit>> {
        let ray = [origin[0], origin[1], dir[0], dir[1]];
        let mut hit = sfx_ray_hit::default();
        check(self.0, unsafe { sfx_query_rays(self.0, 1, ray.as_ptr(), &max_t, mask, &mut hit) })?;
        Ok(if hit.body == SFX_NO_BODY { None } else { Some(hit) })
    }

    /// Bodies whose collider's bounding box overlaps `[lo, hi]`.
    pub fn query_aabb(&mut self, lo: [f32; 2], hi: [f32; 2], mask: u32) -> Result<Vec<u32>> {
        let b = [lo[0], lo[1], hi[0], hi[1]];
        let mut hits = vec![0u64; 64];
        loop {
            let mut n: usize = 0;
            let rc = unsafe { sfx_query_aabbs(self.0, 1, b.as_ptr(), mask, hits.as_mut_ptr(), hits.len(), &mut n) };
            if rc == SFX_ERR_CAPACITY && n > hits.len() {
                hits.resize(n, 0);
                continue;
            }
            check(self.0, rc)?;
            hits.truncate(n);
            return Ok(hits.into_iter().map(|h| h as u32).collect());
        }
    }

    pub fn stats(&mut self) -> Result<sfx_frame_stats> {
        let mut st = std::mem::MaybeUninit::<sfx_frame_stats>::zeroed();
        check(self.0, unsafe { sfx_stats(self.0, st.as_mut_ptr()) })?;
        Ok(unsafe { st.assume_init() })
    }

    pub fn raw(&mut self) -> *mut sfx_world {
        self.0
    }
}

impl Drop for World {
    fn drop(&mut self) {
        unsafe { sfx_destroy(self.0) }
    }
}
