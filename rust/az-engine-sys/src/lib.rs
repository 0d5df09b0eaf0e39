//! Raw bindings of include/az_engine.h (EVERY exported entry point, checked against the header by
//! tests/test_abi_cpu.py) plus a safe `Mi355xNNet` with the reference's `NNet` trait surface (src/nnet.rs:35-45:
//! `new`, `predict`, `train`), an `Mi355xMcts` with `AsyncMcts`'s (`default`, `from_state`, `get_action_prob`,
//! src/async_mcts.rs:27-115), and helpers that replace the self-play fan-out (src/coach.rs:241-272) and the arena gate
//! (src/coach.rs:333-390) with one engine call each.  NOT compiled in this repository (no Rust toolchain here).
#![allow(non_camel_case_types)]
use std::ffi::{CStr, CString};
use std::os::raw::{c_char, c_int};
use std::path::Path;

use ndarray::{Array1, Array2, ArrayView1, ArrayView2, ArrayViewD};

#[repr(C)] pub struct az_engine { _p: [u8; 0] }
#[repr(C)] pub struct az_tree { _p: [u8; 0] }

#[repr(C)] #[derive(Default, Clone, Copy)]
pub struct az_config { pub device: i32, pub max_batch: i32, pub net_channels: i32, pub profile: i32, pub game: i32 }

#[repr(C)] #[derive(Default, Clone, Copy)]
pub struct az_stats {
    pub games: u64, pub moves: u64, pub simulations: u64, pub expansions: u64, pub leaf_evals: u64,
    pub link_hits: u64, pub terminal_hits: u64, pub depth_sum: u64, pub samples: u64, pub net_launches: u64,
    pub net_conv2_ms: f64, pub net_conv2_flops: f64, pub net_total_ms: f64, pub net_total_flops: f64,
    pub tree_ms: f64, pub tree_bytes: f64, pub device_ms: f64,
    pub leaf_rows_requested: u64, pub leaf_rows_executed: u64, pub eval_cache_hits: u64, pub eval_batch_dups: u64,
    pub eval_cache_inserts: u64, pub net_conv3_ms: f64, pub net_conv3_flops: f64, pub net_conv2_bytes: f64,
    pub tree_launches: u64, pub tree_launches_timed: u64, pub tree_arena_allocs: u64,
    pub net_conv4_ms: f64, pub net_conv4_flops: f64, pub net_fc_ms: f64, pub net_fc_flops: f64, pub net_rows_timed: f64,
    pub abandoned_sims: u64, pub net_conv3_image_rows: u64, pub net_conv3_image_launches: u64,
}

#[repr(C)] #[derive(Clone, Copy)]
pub struct az_selfplay_params {
    pub n_games: i32, pub concurrent: i32, pub num_sims: i32, pub temp_threshold: i32,
    pub max_depth: i32, pub cpuct: i32, pub model_id: i32, pub symmetries: i32,
    pub reserve: u64, pub seed: u64, pub first_game_id: u64, pub record_evals: i32, pub num_sim_threads: i32,
}
#[repr(C)]
pub struct az_samples {
    pub capacity: i64, pub count: i64, pub states: *mut u64, pub boards: *mut f32,
    pub pis: *mut f32, pub zs: *mut f32, pub game_len: *mut i32, pub moves: *mut u8,
}
#[repr(C)] #[derive(Clone, Copy)]
pub struct az_arena_params {
    pub num_games: i32, pub num_sims: i32, pub max_depth: i32, pub cpuct: i32,
    pub new_model_id: i32, pub old_model_id: i32, pub reserve: u64, pub seed: u64,
    pub first_game: i32, pub total_games: i32, pub record_evals: i32, pub num_sim_threads: i32,
    pub use_start_board: i32, pub allreduce_wld: i32, pub start_board: [u64; 2],
}

extern "C" {
    // ---- lifecycle
    pub fn az_create(cfg: *const az_config, out: *mut *mut az_engine) -> c_int;
    pub fn az_destroy(e: *mut az_engine);
    pub fn az_last_error(e: *const az_engine) -> *const c_char;
    pub fn az_set_option(e: *mut az_engine, key: *const c_char, value: i64) -> c_int;
    pub fn az_get_stats(e: *mut az_engine, out: *mut az_stats) -> c_int;
    pub fn az_reset_stats(e: *mut az_engine) -> c_int;
    // ---- NNet trait, src/nnet.rs:35-45
    pub fn az_net_set_kind(e: *mut az_engine, model_id: i32, kind: c_int, salt: u64) -> c_int;
    pub fn az_net_free(e: *mut az_engine, model_id: i32) -> c_int;
    pub fn az_net_init_random(e: *mut az_engine, model_id: i32, seed: u64) -> c_int;
    pub fn az_net_load(e: *mut az_engine, model_id: i32, path: *const c_char) -> c_int;
    pub fn az_net_save(e: *mut az_engine, model_id: i32, path: *const c_char) -> c_int;
    pub fn az_net_param_count(e: *const az_engine) -> i64;
    pub fn az_net_set_params(e: *mut az_engine, model_id: i32, params: *const f32, n: i64) -> c_int;
    pub fn az_net_get_params(e: *mut az_engine, model_id: i32, params: *mut f32, n: i64) -> c_int;
    pub fn az_net_predict(e: *mut az_engine, model_id: i32, boards: *const f32, B: i32, pi: *mut f32, v: *mut f32) -> c_int;
    pub fn az_net_predict_states(e: *mut az_engine, model_id: i32, states: *const u64, B: i32, pi: *mut f32, v: *mut f32) -> c_int;
    pub fn az_net_train(e: *mut az_engine, prev_id: i32, id: i32, boards: *const f32, pis: *const f32, vs: *const f32, n: i64) -> c_int;
    pub fn az_net_train_history(e: *const az_engine, out: *mut f32, cap_epochs: i32) -> i32;
    pub fn az_net_train_begin(e: *mut az_engine, prev_id: i32) -> c_int;
    pub fn az_net_train_step(e: *mut az_engine, boards: *const f32, pis: *const f32, vs: *const f32, b: i32, mask_seed: u64,
                             apply: i32, loss_out: *mut f32, grads_out: *mut f32) -> c_int;
    pub fn az_net_train_end(e: *mut az_engine, model_id: i32) -> c_int;
    // ---- AsyncMcts, src/async_mcts.rs:14-115
    pub fn az_tree_create(e: *mut az_engine, n_games: i32, reserve: u64, num_sims: i32, num_threads: i32, max_depth: i32,
                          model_id: i32, cpuct: i32, out: *mut *mut az_tree) -> c_int;
    pub fn az_tree_destroy(t: *mut az_tree);
    pub fn az_tree_reset(t: *mut az_tree, root_states: *const u64) -> c_int;
    pub fn az_tree_get_action_prob(t: *mut az_tree, states: *const u64, temp: f32, seed: u64, first_game_id: u64,
                                   pi: *mut f32, counts: *mut u16, q: *mut f32) -> c_int;
    pub fn az_tree_record_evals(t: *mut az_tree, cap: i32) -> c_int;
    pub fn az_tree_get_evals(t: *mut az_tree, rec_count: *mut i32, states: *mut u64, pis: *mut f32, vs: *mut f32) -> c_int;
    pub fn az_tree_node_counts(t: *mut az_tree, out: *mut u32) -> c_int;
    // ---- Coach::execute_episode x many, src/coach.rs:104-157; arena::play_games, src/arena.rs:62-99
    pub fn az_selfplay(e: *mut az_engine, p: *const az_selfplay_params, out: *mut az_samples) -> c_int;
    // the same as a session: the slots stay full across the calls that fetch the episodes (no drain per chunk)
    pub fn az_selfplay_begin(e: *mut az_engine, p: *const az_selfplay_params) -> c_int;
    pub fn az_selfplay_next(e: *mut az_engine, n_games: i32, out: *mut az_samples) -> c_int;
    pub fn az_selfplay_end(e: *mut az_engine) -> c_int;
    pub fn az_selfplay_get_evals(e: *mut az_engine, rec_count: *mut i32, states: *mut u64, pis: *mut f32, vs: *mut f32) -> c_int;
    pub fn az_arena(e: *mut az_engine, p: *const az_arena_params, out_wld: *mut u64, results: *mut i8) -> c_int;
    pub fn az_arena_get_evals(e: *mut az_engine, which: i32, rec_count: *mut i32, states: *mut u64, pis: *mut f32, vs: *mut f32) -> c_int;
    pub fn az_arena_get_moves(e: *mut az_engine, game_len: *mut i32, moves: *mut u8) -> c_int;
    // ---- the collective of the sharded Coach loop (one process per GPU; RCCL on the engine's stream)
    pub fn az_comm_unique_id(e: *mut az_engine, id: *mut u8) -> c_int;
    pub fn az_comm_init(e: *mut az_engine, rank: i32, world: i32, id: *const u8) -> c_int;
    pub fn az_comm_destroy(e: *mut az_engine) -> c_int;
    pub fn az_gather_samples(e: *mut az_engine, local: *const az_samples, dst_rank: i32, gathered: *mut az_samples, counts_out: *mut i64) -> c_int;
    pub fn az_allreduce_u64(e: *mut az_engine, values: *mut u64, n: i32) -> c_int;
}

/// The reference panics on every error (unwrap/assert!); keep that behaviour.
pub fn check(e: *const az_engine, rc: c_int) {
    if rc != 0 {
        let msg = unsafe { CStr::from_ptr(az_last_error(e)) }.to_string_lossy().into_owned();
        panic!("az_engine status {}: {}", rc, msg);
    }
}

/// `impl NNet` (src/nnet.rs:35-45) over the engine; drop-in for `PythonNNet` (examples/utils/python_nnet.rs).
pub struct Mi355xNNet { pub e: *mut az_engine }

impl Mi355xNNet {
    pub fn new<P: AsRef<Path>>(checkpoint: P) -> Self {                               // NNet::new, src/nnet.rs:36
        let mut e = std::ptr::null_mut();
        assert_eq!(unsafe { az_create(&az_config::default(), &mut e) }, 0, "az_create failed");
        let p = CString::new(checkpoint.as_ref().join("0.aznet").to_str().unwrap()).unwrap();
        if unsafe { az_net_load(e, 0, p.as_ptr()) } != 0 { check(e, unsafe { az_net_init_random(e, 0, 0) }); }
        Mi355xNNet { e }
    }
    /// NNet::predict(board [B,2,6,7], model_id) -> (pi [B,7], v [B]), src/nnet.rs:40-44
    pub fn predict(&self, board: ArrayViewD<f32>, model_id: usize) -> (Array2<f32>, Array1<f32>) {
        let b = board.shape()[0];
        let x = board.as_standard_layout();
        let (mut pi, mut v) = (Array2::<f32>::zeros((b, 7)), Array1::<f32>::zeros(b));
        check(self.e, unsafe { az_net_predict(self.e, model_id as i32, x.as_ptr(), b as i32, pi.as_mut_ptr(), v.as_mut_ptr()) });
        (pi, v)
    }
    /// NNet::train(examples = (boards [N,2,6,7], pis [N,7], vs [N]), previous_model_id, model_id), src/nnet.rs:38:
    /// the reference's recipe on the device (f32 MFMA trainer), result stored under model_id.
    pub fn train(&mut self, boards: ArrayViewD<f32>, pis: ArrayView2<f32>, vs: ArrayView1<f32>, previous_model_id: usize, model_id: usize) {
        let (b, p, v) = (boards.as_standard_layout(), pis.as_standard_layout(), vs.as_standard_layout());
        check(self.e, unsafe { az_net_train(self.e, previous_model_id as i32, model_id as i32, b.as_ptr(), p.as_ptr(), v.as_ptr(), v.len() as i64) });
    }
    /// Drop a superseded model id (Coach::learn moves to model_id + 1 per accepted iteration, src/coach.rs:383-390).
    pub fn free(&mut self, model_id: usize) { check(self.e, unsafe { az_net_free(self.e, model_id as i32) }); }
}
impl Drop for Mi355xNNet { fn drop(&mut self) { unsafe { az_destroy(self.e) } } }

/// n x `AsyncMcts<ConnectFourGame>` (src/async_mcts.rs:14-115) as one tree batch on the device.
pub struct Mi355xMcts { pub e: *mut az_engine, pub t: *mut az_tree, pub n_games: usize }

impl Mi355xMcts {
    /// AsyncMcts::default(reserve_space, num_sims, num_threads, max_depth, model_id, cpuct, ..), src/async_mcts.rs:27-48;
    /// num_threads > 1 = several simulations in flight per tree as a deterministic lock-step schedule (az_engine.h)
    pub fn default(e: *mut az_engine, n_games: usize, reserve_space: usize, num_sims: usize, num_threads: usize, max_depth: usize, model_id: usize, cpuct: i32) -> Self {
        let mut t = std::ptr::null_mut();
        check(e, unsafe { az_tree_create(e, n_games as i32, reserve_space as u64, num_sims as i32, num_threads as i32, max_depth as i32, model_id as i32, cpuct, &mut t) });
        Mi355xMcts { e, t, n_games }
    }
    /// AsyncMcts::from_state(s, ..), src/async_mcts.rs:50-72: every tree re-rooted at its canonical bitboards [n_games, 2]
    pub fn from_state(e: *mut az_engine, root_states: &[u64], reserve_space: usize, num_sims: usize, num_threads: usize, max_depth: usize, model_id: usize, cpuct: i32) -> Self {
        let m = Self::default(e, root_states.len() / 2, reserve_space, num_sims, num_threads, max_depth, model_id, cpuct);
        check(e, unsafe { az_tree_reset(m.t, root_states.as_ptr()) });
        m
    }
    /// get_action_prob(&self, s, temp, episode_id, rng) for every tree, src/async_mcts.rs:74-115: pi [n_games, 7]
    pub fn get_action_prob(&self, states: &[u64], temp: f32, seed: u64, first_game_id: u64) -> Array2<f32> {
        let mut pi = Array2::<f32>::zeros((self.n_games, 7));
        check(self.e, unsafe { az_tree_get_action_prob(self.t, states.as_ptr(), temp, seed, first_game_id, pi.as_mut_ptr(),
                                                       std::ptr::null_mut(), std::ptr::null_mut()) });
        pi
    }
    /// NodeStore::len per tree, src/node.rs:372-374
    pub fn node_counts(&self) -> Vec<u32> {
        let mut out = vec![0u32; self.n_games];
        check(self.e, unsafe { az_tree_node_counts(self.t, out.as_mut_ptr()) });
        out
    }
}
impl Drop for Mi355xMcts { fn drop(&mut self) { unsafe { az_tree_destroy(self.t) } } }

/// Counters since az_create / the last reset (SURVEY.md 8b "Introspection").
pub fn stats(e: *mut az_engine) -> az_stats {
    let mut s = az_stats::default();
    check(e, unsafe { az_get_stats(e, &mut s) });
    s
}

/// Replaces the rayon fan-out of `execute_episode` (src/coach.rs:241-272): returns (boards [N,2,6,7], pis [N,7], vs [N]).
pub fn self_play(e: *mut az_engine, p: &az_selfplay_params) -> (Vec<f32>, Vec<f32>, Vec<f32>) {
    let cap = p.n_games as usize * 42 * if p.symmetries != 0 { 2 } else { 1 };
    let (mut boards, mut pis, mut zs) = (vec![0f32; cap * 84], vec![0f32; cap * 7], vec![0f32; cap]);
    let mut out = az_samples { capacity: cap as i64, count: 0, states: std::ptr::null_mut(), boards: boards.as_mut_ptr(),
                               pis: pis.as_mut_ptr(), zs: zs.as_mut_ptr(), game_len: std::ptr::null_mut(), moves: std::ptr::null_mut() };
    check(e, unsafe { az_selfplay(e, p, &mut out) });
    let n = out.count as usize;
    boards.truncate(n * 84); pis.truncate(n * 7); zs.truncate(n);
    (boards, pis, zs)
}

/// The ONE exchange of a sharded episode batch (each rank played `p.first_game_id ..` of the global episode ids): this rank's
/// (states [n,2], pis [n,7], zs [n]) go to `dst_rank`, which gets everybody's tuples in rank order (else empty vectors);
/// `dst_rank = -1`: every rank receives everything.  `global_episodes` = the number of episodes of the WHOLE batch over all ranks:
/// a receiving rank sizes its buffers from the hard bound 42 plies per episode (uneven shards make any bound derived from the
/// local count too small; the library checks the capacity on every rank before anything is posted, so a wrong bound is an error
/// on all ranks, never a hang).
pub fn gather_samples(e: *mut az_engine, states: &mut [u64], pis: &mut [f32], zs: &mut [f32], rank: i32, world: i32, dst_rank: i32,
                      global_episodes: usize) -> (Vec<u64>, Vec<f32>, Vec<f32>) {
    let n = zs.len();
    let local = az_samples { capacity: n as i64, count: n as i64, states: states.as_mut_ptr(), boards: std::ptr::null_mut(),
                             pis: pis.as_mut_ptr(), zs: zs.as_mut_ptr(), game_len: std::ptr::null_mut(), moves: std::ptr::null_mut() };
    let mut counts = vec![0i64; world as usize];
    if dst_rank >= 0 && rank != dst_rank {
        check(e, unsafe { az_gather_samples(e, &local, dst_rank, std::ptr::null_mut(), counts.as_mut_ptr()) });
        return (vec![], vec![], vec![]);
    }
    let cap = global_episodes * 42;
    let (mut gs, mut gp, mut gz) = (vec![0u64; cap * 2], vec![0f32; cap * 7], vec![0f32; cap]);
    let mut g = az_samples { capacity: cap as i64, count: 0, states: gs.as_mut_ptr(), boards: std::ptr::null_mut(),
                             pis: gp.as_mut_ptr(), zs: gz.as_mut_ptr(), game_len: std::ptr::null_mut(), moves: std::ptr::null_mut() };
    check(e, unsafe { az_gather_samples(e, &local, dst_rank, &mut g, counts.as_mut_ptr()) });
    let m = g.count as usize;
    gs.truncate(m * 2); gp.truncate(m * 7); gz.truncate(m);
    (gs, gp, gz)
}

/// Replaces `play_games` + the tally of src/coach.rs:365-381: (nwins, pwins, draws) for the new model.
pub fn arena(e: *mut az_engine, p: &az_arena_params) -> (u64, u64, u64) {
    let mut wld = [0u64; 3];
    check(e, unsafe { az_arena(e, p, wld.as_mut_ptr(), std::ptr::null_mut()) });
    (wld[0], wld[1], wld[2])
}
