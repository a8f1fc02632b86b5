/* Debugging aid (never linked into the product): LD_PRELOAD=libpoison_malloc.so
 *  - fills every hipMalloc'ed block with 0xFF bytes (a NaN as float or double, -1 as int) before handing it out, so that a read of
 *    memory the library never initialised shows up as NaN in the results instead of depending on what the previous owner of the
 *    pages left there (TSX_POISON_BYTE=<n> picks another byte);
 *  - TSX_GUARD=1: puts a 4 KiB guard zone of 0xA5 in front of and behind every block and checks both at hipFree: a kernel that
 *    writes a few elements past its buffer is reported (stderr: the block's size, its allocation number, which side, the first
 *    damaged offset; appended to the file TSX_GUARD_LOG names, if set) wherever the neighbour in memory happens to be.
 *  - TSX_QUARANTINE=1 (with TSX_GUARD=1): freed blocks are never handed out again but filled with 0xEE and watched for writes
 *    through stale pointers (blocks up to 64 MiB).
 *   gcc -O2 -shared -fPIC -o libpoison_malloc.so poison_malloc.c -ldl
 *   LD_PRELOAD=$PWD/libpoison_malloc.so python -m pytest tests -m gpu -k ... */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef int (*malloc_fn)(void **, size_t);
typedef int (*free_fn)(void *);
typedef int (*memset_fn)(void *, int, size_t);
typedef int (*memcpy_fn)(void *, const void *, size_t, int);
typedef int (*sync_fn)(void);

#define GUARD 4096
#define MAXB 65536
static struct { char *user; size_t n; long id; } blocks[MAXB];
static long nalloc;
static malloc_fn real_malloc;
static free_fn real_free;
static memset_fn real_set;
static memcpy_fn real_copy;
static sync_fn real_sync;
static int byte = 0xFF, guard = 0;

#define MAXQ 16384
static struct { char *base; size_t n; long id; int reported; } quar[MAXQ];
static int nquar, quarantine;
static void check_quarantine(void) {
  static unsigned char *h;
  static size_t hcap;
  for (int q = 0; q < nquar; ++q) {
    if (quar[q].reported) continue;
    if (hcap < quar[q].n) {
      free(h);
      h = (unsigned char *)malloc(quar[q].n);
      hcap = quar[q].n;
    }
    if (!h || real_copy(h, quar[q].base, quar[q].n, 2) != 0) continue;
    for (size_t b = 0; b < quar[q].n; ++b)
      if (h[b] != 0xEE) {
        FILE *out = stderr;
        const char *lf = getenv("TSX_GUARD_LOG");
        if (lf) out = fopen(lf, "a");
        if (!out) out = stderr;
        fprintf(out, "TSX_QUARANTINE: freed block #%ld of %zu bytes was written after hipFree, first at offset %ld\n", quar[q].id, quar[q].n - 2 * GUARD,
                (long)b - GUARD);
        if (out != stderr) fclose(out);
        quar[q].reported = 1;
        break;
      }
  }
}

static void init(void) {
  if (real_malloc) return;
  real_malloc = (malloc_fn)dlsym(RTLD_NEXT, "hipMalloc");
  real_free = (free_fn)dlsym(RTLD_NEXT, "hipFree");
  real_set = (memset_fn)dlsym(RTLD_NEXT, "hipMemset");
  real_copy = (memcpy_fn)dlsym(RTLD_NEXT, "hipMemcpy");
  real_sync = (sync_fn)dlsym(RTLD_NEXT, "hipDeviceSynchronize");
  const char *e = getenv("TSX_POISON_BYTE");
  if (e) byte = atoi(e);
  e = getenv("TSX_GUARD");
  guard = e && atoi(e) != 0;
  e = getenv("TSX_QUARANTINE");
  quarantine = guard && e && atoi(e) != 0;
}

int hipMalloc(void **p, size_t n) {
  init();
  if (!guard) {
    int rc = real_malloc(p, n);
    if (rc == 0 && n) {
      real_set(*p, byte, n);
      real_sync();
    }
    return rc;
  }
  char *base = NULL;
  int rc = real_malloc((void **)&base, n + 2 * GUARD);
  if (rc) return rc;
  real_set(base, 0xA5, n + 2 * GUARD);
  if (n) real_set(base + GUARD, byte, n);
  real_sync();
  *p = base + GUARD;
  ++nalloc;
  for (int q = 0; q < MAXB; ++q)
    if (!blocks[q].user) {
      blocks[q].user = base + GUARD;
      blocks[q].n = n;
      blocks[q].id = nalloc;
      break;
    }
  return 0;
}

int hipFree(void *p) {
  init();
  if (!guard || !p) return real_free(p);
  for (int q = 0; q < MAXB; ++q)
    if (blocks[q].user == (char *)p) {
      static unsigned char h[GUARD];
      real_sync();
      for (int side = 0; side < 2; ++side) {
        const char *src = side ? blocks[q].user + blocks[q].n : blocks[q].user - GUARD;
        if (real_copy(h, src, GUARD, 2 /* hipMemcpyDeviceToHost */) != 0) continue;
        for (int b = 0; b < GUARD; ++b)
          if (h[b] != 0xA5) {
            FILE *out = stderr;
            const char *lf = getenv("TSX_GUARD_LOG");  /* (pytest swallows the workers' stderr of a passing test) */
            if (lf) out = fopen(lf, "a");
            if (!out) out = stderr;
            fprintf(out, "TSX_GUARD: block #%ld of %zu bytes: the guard zone %s it is damaged from offset %d (%s the block's %s)\n", blocks[q].id,
                    blocks[q].n, side ? "behind" : "in front of", side ? b : b - GUARD, side ? "past" : "before", side ? "end" : "start");
            if (out != stderr) fclose(out);
            break;
          }
      }
      blocks[q].user = NULL;
      if (quarantine && nquar < MAXQ && blocks[q].n + 2 * GUARD <= (64u << 20)) {
        /* TSX_QUARANTINE=1: the block is not given back but filled with 0xEE and watched: a kernel that still writes through a
         * stale pointer damages the pattern (reported at a later hipFree), one that still reads through it computes with NaN */
        real_set((char *)p - GUARD, 0xEE, blocks[q].n + 2 * GUARD);
        real_sync();
        quar[nquar].base = (char *)p - GUARD;
        quar[nquar].n = blocks[q].n + 2 * GUARD;
        quar[nquar].id = blocks[q].id;
        ++nquar;
        check_quarantine();
        return 0;
      }
      check_quarantine();
      return real_free((char *)p - GUARD);
    }
  return real_free(p);  /* not ours (allocated before the shim was active) */
}
