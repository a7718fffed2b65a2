// handle_access.hpp — what the multi-GPU layer (particles_comm.hip) needs from a
// handle beyond the public C ABI.  Internal to the library.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>

#include <string>

#include "particles_hip.h"

namespace pcl_internal {

int fail(int code, const std::string& message);          // sets pcl_last_error(), returns code
int make_current(const pcl_compute* h);                   // hipSetDevice(h's device)
int device_of(const pcl_compute* h);
hipStream_t stream_of(const pcl_compute* h);
// Sizes the handle's field for n_cells cells (as a whole-field replace does) and returns its
// device address, so that a collective can deliver the cells straight into it.
int field_for_write(pcl_compute* h, size_t n_cells, void** dev_field);
void* field_of(pcl_compute* h);                          // for writing: the packed copy is rebuilt before the next step
size_t cells_of(const pcl_compute* h);
// Reduces the diagnostics of the latest state on the handle's stream (unless a fused per-step
// result is current) and returns the device address of the 48-byte record.
int diagnostics_on_device(pcl_compute* h, const void** dev_record);
// Where the handle's reductions write the 48-byte record from now on (device memory of the caller's;
// null: the handle's own record again).  The multi-GPU layer alternates between two records of its own,
// so that a collective may still read one step's record while the next step writes the other.
void set_diag_target(pcl_compute* h, void* dev_record);
bool pair_term_active(const pcl_compute* h);              // the parameters' pair term is not provably zero
bool step_diagnostics_enabled(const pcl_compute* h);
// Goes up with every call that replaces the particle set (upload, SoA upload, generate), whether or
// not the count changed: ranks that make the same calls see it change between the same two steps.
uint64_t particles_epoch(const pcl_compute* h);
// One slot for the multi-GPU layer's per-handle state; `release` runs in pcl_destroy, `sync` (may be
// null) in every call that waits for the handle's stream, before that wait: the layer's own stream
// carries work that the handle's stream does not wait for.
void set_comm(pcl_compute* h, void* comm, void (*release)(void*), int (*sync)(void*) = nullptr);
void* comm_of(const pcl_compute* h);

}  // namespace pcl_internal
