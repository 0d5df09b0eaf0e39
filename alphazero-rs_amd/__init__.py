"""MI355X-native AlphaZero self-play engine for the async_mcts + arena hot path of
AnimatedRNG/alphazero-rs.  The product is the HIP library `libaz_engine.so`
(C ABI in include/az_engine.h); this package is the thin host-side binding:

  build    -- compiles the HIP sources for gfx950 (no GPU needed)
  engine   -- ctypes mirror of the C ABI (Engine, TreeBatch)
  dist     -- one-process-per-GPU sharding + the RCCL gather of (s, pi, z)

There is no CPU fallback: importing `engine` without the built library raises.
"""
__all__ = ["build", "engine", "dist"]
