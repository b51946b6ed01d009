// gpu_devices.h -- which GPUs a Gpu*Index spreads over (shared by gpu_dcthashindex.h and gpu_indexes.h).
#pragma once

#include <stdint.h>

#include "cbird_hip.h"

/// Which GPUs an index spreads over.  cbird is one process that registers each index once (src/engine.cpp:38-45), so
/// the shards of a multi-GPU index live INSIDE the index object: `mask` has bit d set for every HIP device d that takes
/// a share of the rows; `shardsPerDevice` > 1 cuts each device's share into logical shards (own streams).  The search
/// results do not depend on either (cbird_amd/csrc/sharded.hip).
struct GpuDeviceSet {
  uint32_t mask = 1u;
  int shardsPerDevice = 1;
  /// every usable gfx950 device of the node, e.g. 0xff on an 8 x MI355X box (probed per ordinal: a node that lists some
  /// other device first gives a mask with holes, not the wrong devices); no usable device: mask 1, and the index
  /// reports the missing device when it is created
  static GpuDeviceSet all() {
    const uint32_t m = cbh_usable_device_mask();
    return GpuDeviceSet{m ? m : 1u, 1};
  }
  bool single() const { return (mask & (mask - 1)) == 0 && shardsPerDevice <= 1; }
  int first() const { return mask ? __builtin_ctz(mask) : 0; }
};
