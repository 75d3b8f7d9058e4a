/* Compiled by tests/test_host_logic.py with a plain C compiler: the boundary header must be valid C99 (no C++,
 * no torch/HIP types) and every entry point must be callable through a C function pointer of its declared type. */
#define _POSIX_C_SOURCE 200809L
#include <stddef.h>
#include <stdio.h>
#include <dlfcn.h>

#include "umetrack_hip.h"

#define CHECK_SYM(name)                                                  \
  do {                                                                   \
    void* p_ = dlsym(lib, #name);                                        \
    if (!p_) { fprintf(stderr, "missing %s\n", #name); return 2; }       \
    *(void**)(&fn_##name) = p_; /* the POSIX dlsym idiom */              \
    (void)fn_##name;                                                     \
  } while (0)

typedef size_t (*type_ut_weight_blob_floats)(void);
typedef const char* (*type_ut_last_error)(ut_handle);
typedef int (*type_ut_fk)(ut_handle, const float*, int, const float*, int, const float*, int, const int64_t*, float, int,
                          float*, void*);
typedef int (*type_ut_resample_homography)(ut_handle, const void*, int, int, int, int, const float*, int, int, float*, void*);

int main(int argc, char** argv) {
  type_ut_weight_blob_floats fn_ut_weight_blob_floats;
  type_ut_last_error fn_ut_last_error;
  type_ut_fk fn_ut_fk;
  type_ut_resample_homography fn_ut_resample_homography;
  void* lib;
  if (argc < 2) return 1;
  lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!lib) { fprintf(stderr, "%s\n", dlerror()); return 3; }
  CHECK_SYM(ut_weight_blob_floats);
  CHECK_SYM(ut_last_error);
  CHECK_SYM(ut_fk);
  CHECK_SYM(ut_resample_homography);
  /* no GPU needed for these: sizes and argument validation happen on the host */
  printf("%lu\n", (unsigned long)fn_ut_weight_blob_floats());
  if (fn_ut_fk(NULL, NULL, 1, NULL, 22, NULL, 16, NULL, 1.0f, 4, NULL, NULL) == UT_OK) return 4;
  if (fn_ut_resample_homography(NULL, NULL, 0, 1, 4, 4, NULL, 4, 4, NULL, NULL) == UT_OK) return 5;
  if (!fn_ut_last_error(NULL) || !fn_ut_last_error(NULL)[0]) return 6;
  return 0;
}
