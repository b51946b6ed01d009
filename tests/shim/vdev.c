/* TEST DOUBLE (LD_PRELOAD): presents the ONE physical GPU of a box as CBH_VDEV devices, so that the branches of
 * cbird_amd/csrc/sharded.hip that only run with more than one device ordinal -- DeviceGuard switching, one arena / stream /
 * workspace set per device, needle replication and the exchange by hipMemcpyPeerAsync, cross-device event waits, the
 * device-mask plumbing of cbh_*_create_sharded and GpuDeviceSet::all() -- execute on this pool, which never has two GPUs
 * in a box.  What it cannot show is anything physical: real peer mappings, xGMI, RCCL between devices (RCCL refuses two
 * ranks on one GPU; tests/test_virtual_devices.py runs the collective shape with "fault_rccl", i.e. the fall-back).
 *
 * Every virtual ordinal maps to physical device 0.  The current ordinal is per thread, as HIP's is.  Only the runtime
 * entry points that take or return a device ordinal are interposed; everything else goes to libamdhip64 untouched.
 * Not product code: nothing under cbird_amd/ knows this file exists.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef int hipError_t;         /* enum in the real header; same ABI */
typedef void* hipStream_t;
typedef void* hipMemPool_t;
enum { kSuccess = 0, kInvalidValue = 1, kInvalidDevice = 101 };
enum { kMemcpyDeviceToDevice = 3 };

static __thread int t_dev = 0;
static int g_n = -1;
/* counters a test can read back (vdev_stat) */
static long g_set_nonzero = 0, g_peer_copies = 0, g_peer_enables = 0;

static int vcount(void) {
  if (g_n < 0) {
    const char* e = getenv("CBH_VDEV");
    int n = e ? atoi(e) : 2;
    g_n = n < 1 ? 1 : (n > 16 ? 16 : n);
  }
  return g_n;
}

#define REAL(name, ...)                                                          \
  static hipError_t (*real)(__VA_ARGS__) = NULL;                                 \
  if (!real) real = (hipError_t(*)(__VA_ARGS__))dlsym(RTLD_NEXT, name);          \
  if (!real) {                                                        \
    fprintf(stderr, "vdev shim: %s not found in the runtime\n", name); \
    abort();                                                          \
  }

static int bad(int d) { return d < 0 || d >= vcount(); }

hipError_t hipGetDeviceCount(int* n) {
  REAL("hipGetDeviceCount", int*);
  int phys = 0;
  hipError_t e = real(&phys);
  if (e != kSuccess) return e;
  if (n) *n = phys >= 1 ? vcount() : 0;
  return kSuccess;
}

hipError_t hipSetDevice(int d) {
  REAL("hipSetDevice", int);
  if (bad(d)) return kInvalidDevice;
  hipError_t e = real(0);
  if (e == kSuccess) {
    t_dev = d;
    if (d) __sync_fetch_and_add(&g_set_nonzero, 1);
  }
  return e;
}

hipError_t hipGetDevice(int* d) {
  REAL("hipGetDevice", int*);
  int phys = 0;
  hipError_t e = real(&phys);  /* (keeps the runtime's lazy initialisation where it was) */
  if (e != kSuccess) return e;
  if (d) *d = t_dev;
  return kSuccess;
}

/* hipGetDeviceProperties is a macro for this symbol since ROCm 6 */
hipError_t hipGetDevicePropertiesR0600(void* prop, int d) {
  REAL("hipGetDevicePropertiesR0600", void*, int);
  if (bad(d)) return kInvalidDevice;
  return real(prop, 0);
}

hipError_t hipDeviceGetAttribute(int* v, int attr, int d) {
  REAL("hipDeviceGetAttribute", int*, int, int);
  if (bad(d)) return kInvalidDevice;
  return real(v, attr, 0);
}

hipError_t hipDeviceGetPCIBusId(char* s, int len, int d) {
  REAL("hipDeviceGetPCIBusId", char*, int, int);
  if (bad(d)) return kInvalidDevice;
  return real(s, len, 0);
}

hipError_t hipDeviceGetName(char* s, int len, int d) {
  REAL("hipDeviceGetName", char*, int, int);
  if (bad(d)) return kInvalidDevice;
  return real(s, len, 0);
}

hipError_t hipDeviceTotalMem(size_t* b, int d) {
  REAL("hipDeviceTotalMem", size_t*, int);
  if (bad(d)) return kInvalidDevice;
  return real(b, 0);
}

hipError_t hipDevicePrimaryCtxGetState(int d, unsigned* flags, int* active) {
  REAL("hipDevicePrimaryCtxGetState", int, unsigned*, int*);
  if (bad(d)) return kInvalidDevice;
  return real(0, flags, active);
}

hipError_t hipDeviceGetP2PAttribute(int* v, int attr, int a, int b) {
  (void)attr;
  if (bad(a) || bad(b)) return kInvalidDevice;
  if (v) *v = 1;
  return kSuccess;
}

hipError_t hipDeviceCanAccessPeer(int* can, int a, int b) {
  if (bad(a) || bad(b)) return kInvalidDevice;
  if (can) *can = a != b;
  return kSuccess;
}

hipError_t hipDeviceEnablePeerAccess(int peer, unsigned flags) {
  (void)flags;
  if (bad(peer) || peer == t_dev) return kInvalidDevice;
  __sync_fetch_and_add(&g_peer_enables, 1);
  return kSuccess;
}

hipError_t hipMemcpyPeer(void* dst, int dd, const void* src, int sd, size_t n) {
  REAL("hipMemcpy", void*, const void*, size_t, int);
  if (bad(dd) || bad(sd)) return kInvalidDevice;
  __sync_fetch_and_add(&g_peer_copies, 1);
  return real(dst, src, n, kMemcpyDeviceToDevice);
}

hipError_t hipMemcpyPeerAsync(void* dst, int dd, const void* src, int sd, size_t n, hipStream_t s) {
  REAL("hipMemcpyAsync", void*, const void*, size_t, int, hipStream_t);
  if (bad(dd) || bad(sd)) return kInvalidDevice;
  __sync_fetch_and_add(&g_peer_copies, 1);
  return real(dst, src, n, kMemcpyDeviceToDevice, s);
}

hipError_t hipDeviceGetDefaultMemPool(hipMemPool_t* pool, int d) {
  REAL("hipDeviceGetDefaultMemPool", hipMemPool_t*, int);
  if (bad(d)) return kInvalidDevice;
  return real(pool, 0);
}

hipError_t hipDeviceGetMemPool(hipMemPool_t* pool, int d) {
  REAL("hipDeviceGetMemPool", hipMemPool_t*, int);
  if (bad(d)) return kInvalidDevice;
  return real(pool, 0);
}

/* hipMemPoolProps: { allocType (4), handleTypes (4), location { type (4), id (4) }, ... } -- the location id is the one
 * ordinal in it ("scratch_mode" 1 creates pools; the default arena never does) */
hipError_t hipMemPoolCreate(hipMemPool_t* pool, const void* props) {
  REAL("hipMemPoolCreate", hipMemPool_t*, const void*);
  unsigned char copy[256];
  memcpy(copy, props, 88);  /* sizeof(hipMemPoolProps) = 88, offsetof(location.id) = 12 (ROCm 7 headers) */
  int id;
  memcpy(&id, copy + 12, 4);
  if (bad(id)) return kInvalidDevice;
  id = 0;
  memcpy(copy + 12, &id, 4);
  return real(pool, copy);
}

/* ---- which ordinal a stream / an event was created under, and the two rules CUDA-style runtimes enforce between them:
 * a kernel goes to a stream of the CURRENT device, an event is recorded on a stream of ITS device.  HIP is lenient about
 * the first today; the library must not depend on that (every per-shard step sits inside a DeviceGuard). */
#include <pthread.h>
typedef void* hipEvent_t;
typedef struct { unsigned x, y, z; } dim3_t;
enum { kTab = 8192 };
static struct { void* h; int dev; } g_tab[kTab];
static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
static long g_wrong_launch = 0, g_wrong_record = 0, g_launches = 0;

static void tab_put(void* h, int dev) {
  if (!h) return;
  pthread_mutex_lock(&g_mu);
  size_t i = ((size_t)h >> 4) % kTab, free_at = kTab;
  for (size_t k = 0; k < kTab; ++k, i = (i + 1) % kTab) {
    if (g_tab[i].h == h) { free_at = i; break; }
    if (!g_tab[i].h || g_tab[i].h == (void*)1) { if (free_at == kTab) free_at = i; if (!g_tab[i].h) break; }
  }
  if (free_at != kTab) g_tab[free_at].h = h, g_tab[free_at].dev = dev;
  pthread_mutex_unlock(&g_mu);
}
static int tab_get(void* h, int drop) { /* -1: unknown (the null stream, handles made before the shim saw them) */
  int dev = -1;
  if (!h) return -1;
  pthread_mutex_lock(&g_mu);
  size_t i = ((size_t)h >> 4) % kTab;
  for (size_t k = 0; k < kTab && g_tab[i].h; ++k, i = (i + 1) % kTab)
    if (g_tab[i].h == h) {
      dev = g_tab[i].dev;
      if (drop) g_tab[i].h = (void*)1; /* tombstone */
      break;
    }
  pthread_mutex_unlock(&g_mu);
  return dev;
}

hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags) {
  REAL("hipStreamCreateWithFlags", hipStream_t*, unsigned);
  hipError_t e = real(s, flags);
  if (e == kSuccess && s) tab_put(*s, t_dev);
  return e;
}
hipError_t hipStreamCreate(hipStream_t* s) {
  REAL("hipStreamCreate", hipStream_t*);
  hipError_t e = real(s);
  if (e == kSuccess && s) tab_put(*s, t_dev);
  return e;
}
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned flags, int prio) {
  REAL("hipStreamCreateWithPriority", hipStream_t*, unsigned, int);
  hipError_t e = real(s, flags, prio);
  if (e == kSuccess && s) tab_put(*s, t_dev);
  return e;
}
hipError_t hipStreamDestroy(hipStream_t s) {
  REAL("hipStreamDestroy", hipStream_t);
  (void)tab_get(s, 1);
  return real(s);
}
int hipGetStreamDeviceId(hipStream_t s) {
  const int d = tab_get(s, 0);
  return d < 0 ? t_dev : d;
}
hipError_t hipEventCreate(hipEvent_t* ev) {
  REAL("hipEventCreate", hipEvent_t*);
  hipError_t e = real(ev);
  if (e == kSuccess && ev) tab_put(*ev, t_dev);
  return e;
}
hipError_t hipEventCreateWithFlags(hipEvent_t* ev, unsigned flags) {
  REAL("hipEventCreateWithFlags", hipEvent_t*, unsigned);
  hipError_t e = real(ev, flags);
  if (e == kSuccess && ev) tab_put(*ev, t_dev);
  return e;
}
hipError_t hipEventDestroy(hipEvent_t ev) {
  REAL("hipEventDestroy", hipEvent_t);
  (void)tab_get(ev, 1);
  return real(ev);
}
hipError_t hipEventRecord(hipEvent_t ev, hipStream_t s) {
  REAL("hipEventRecord", hipEvent_t, hipStream_t);
  const int de = tab_get(ev, 0);
  int ds = tab_get(s, 0);
  if (ds < 0) ds = t_dev;
  if (de >= 0 && de != ds) {
    if (__sync_fetch_and_add(&g_wrong_record, 1) == 0)
      fprintf(stderr, "vdev shim: event of ordinal %d recorded on a stream of ordinal %d\n", de, ds);
  }
  return real(ev, s);
}
hipError_t hipLaunchKernel(const void* f, dim3_t grid, dim3_t block, void** args, size_t shmem, hipStream_t s) {
  REAL("hipLaunchKernel", const void*, dim3_t, dim3_t, void**, size_t, hipStream_t);
  const int ds = tab_get(s, 0);
  __sync_fetch_and_add(&g_launches, 1);
  if (ds >= 0 && ds != t_dev) {
    if (__sync_fetch_and_add(&g_wrong_launch, 1) == 0)
      fprintf(stderr, "vdev shim: kernel launched on a stream of ordinal %d while ordinal %d is current\n", ds, t_dev);
  }
  return real(f, grid, block, args, shmem, s);
}

/* what the shim saw, for the test: 0 = hipSetDevice calls with a non-zero ordinal, 1 = peer copies, 2 = peer enables,
 * 3 = the virtual device count, 4 = kernels launched on a stream of a device that was not current, 5 = events recorded
 * on a stream of another device, 6 = kernel launches seen */
long vdev_stat(int which) {
  switch (which) {
    case 0: return g_set_nonzero;
    case 1: return g_peer_copies;
    case 2: return g_peer_enables;
    case 3: return vcount();
    case 4: return g_wrong_launch;
    case 5: return g_wrong_record;
    case 6: return g_launches;
    default: return -1;
  }
}
