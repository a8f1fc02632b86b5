/* Debugging aid (never linked into the product): LD_PRELOAD=libpoison_malloc.so fills every hipMalloc'ed block with 0xFF bytes
 * (a NaN as float or double, -1 as int) before handing it out, so that a read of memory the library never initialised shows up as NaN
 * in the results instead of depending on what the previous owner of the pages left there.
 *   gcc -O2 -shared -fPIC -o libpoison_malloc.so poison_malloc.c -ldl
 *   LD_PRELOAD=$PWD/libpoison_malloc.so python -m pytest tests -m gpu -k ... */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stddef.h>
#include <stdlib.h>

typedef int (*malloc_fn)(void **, size_t);
typedef int (*memset_fn)(void *, int, size_t);
typedef int (*sync_fn)(void);

int hipMalloc(void **p, size_t n) {
  static malloc_fn real;
  static memset_fn set;
  static sync_fn sync;
  static int byte = 0xFF;
  if (!real) {
    real = (malloc_fn)dlsym(RTLD_NEXT, "hipMalloc");
    set = (memset_fn)dlsym(RTLD_NEXT, "hipMemset");
    sync = (sync_fn)dlsym(RTLD_NEXT, "hipDeviceSynchronize");
    const char *e = getenv("TSX_POISON_BYTE");
    if (e) byte = atoi(e);
  }
  int rc = real(p, n);
  if (rc == 0 && n && set) {
    set(*p, byte, n);
    if (sync) sync();
  }
  return rc;
}
