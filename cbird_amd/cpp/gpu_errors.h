// gpu_errors.h -- what a Gpu*Index does with a return code of the C-ABI (shared by gpu_dcthashindex.h and
// gpu_indexes.h).  Include after cbird's index.h: qFatal / qCritical are Qt's logging macros.
#pragma once

#include "cbird_hip.h"

/// What a Gpu*Index does with a return code of the C-ABI.  cbird's Index surface has no error channel.  The reference
/// aborts only where it cannot go on -- an SQL statement failed (SQL_FATAL, src/global.h:82), its arrays could not
/// grow -- and answers missing needle data, an empty index or a corrupt row with a log line and an empty result
/// (src/dcthashindex.cpp:196-205, src/dctfeaturesindex.cpp:145-148).  The adapters follow that split:
///   * a failed ALLOCATION is first treated as transient (a device filled up by another index's cached scratch):
///     CBH_E_NOMEM, or CBH_E_OVERFLOW where it stands for one (entry points that could not grow a result buffer say
///     OVERFLOW; cbh_last_error_code() then says NOMEM).  The cached scratch of the index's own devices goes back to
///     the driver (cbh_trim) and the call runs once more.  An OVERFLOW that no allocation caused is deterministic (the
///     result exceeds what the entry point can hold): running it again would only stall every other thread's streams
///     behind cbh_trim's device synchronisation and scan twice;
///   * a QUERY (find, findIndexData, mediaIds, slice) that still fails logs with qCritical and returns nothing -- a
///     1M-needle -similar loses one needle's matches, not the process;
///   * a MUTATION (load, add, remove) that still fails leaves the index out of step with the database: qFatal, as the
///     reference's own failed allocation does.
namespace gpuidx {
enum Kind { Query, Mutation };
/// mask = the devices the failing index lives on (0: every usable device)
inline void releaseScratch(uint32_t mask = 0) {
  for (uint32_t m = mask ? mask : cbh_usable_device_mask(); m; m &= m - 1) (void)cbh_trim(__builtin_ctz(m), nullptr);
}
template <class Call>
inline bool run(Kind kind, const char* what, Call&& call, uint32_t mask = 0) {
  cbh_clear_error();  // so that the code read below belongs to this call
  int rc = call();
  if (rc == CBH_E_NOMEM || (rc == CBH_E_OVERFLOW && cbh_last_error_code() == CBH_E_NOMEM)) {
    releaseScratch(mask);
    cbh_clear_error();
    rc = call();
  }
  if (rc == CBH_OK) return true;
  if (kind == Mutation) qFatal("%s: %s (%s)", what, cbh_strerror(rc), cbh_last_error());
  qCritical("%s: %s (%s) -- no results for this call", what, cbh_strerror(rc), cbh_last_error());
  return false;
}
}  // namespace gpuidx
