/*
 * libenv.h -- restatement of the gym3 "libenv" C ABI (gym3 0.3.3, LIBENV_VERSION 1).
 *
 * The reference (openai/procgen) does not vendor this header: it is taken from
 * gym3.libenv.get_header_dir() at build time (reference procgen/builder.py:83,
 * procgen/CMakeLists.txt:76).  gym3 is not installed in this image, so the ABI is
 * restated here from the way the reference uses it:
 *   field names / types of libenv_tensortype   -> reference src/vecgame.cpp:213-281
 *   libenv_options / libenv_option              -> reference src/vecoptions.cpp:5,40-57
 *   libenv_buffers (ac/ob/info/rew/first)       -> reference src/vecgame.cpp:74-83
 *   the seven entry points                      -> reference src/vecgame.cpp:42-99
 * The restatement is self-consistency checked: the unmodified reference sources
 * compile and run against this file (oracle/Makefile builds oracle/_ref/libenv.so
 * with -I include).
 */
#pragma once

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LIBENV_VERSION 1
#define LIBENV_MAX_NAME_LEN 128
#define LIBENV_MAX_NDIM 16

#if defined(_WIN32)
#define LIBENV_API __declspec(dllexport)
#else
#define LIBENV_API __attribute__((__visibility__("default")))
#endif

enum libenv_dtype {
    LIBENV_DTYPE_UNUSED = 0,
    LIBENV_DTYPE_UINT8 = 1,
    LIBENV_DTYPE_INT32 = 2,
    LIBENV_DTYPE_FLOAT32 = 3,
};

enum libenv_scalar_type {
    LIBENV_SCALAR_TYPE_UNUSED = 0,
    LIBENV_SCALAR_TYPE_REAL = 1,
    LIBENV_SCALAR_TYPE_DISCRETE = 2,
};

enum libenv_space_name {
    LIBENV_SPACE_UNUSED = 0,
    LIBENV_SPACE_OBSERVATION = 1,
    LIBENV_SPACE_ACTION = 2,
    LIBENV_SPACE_INFO = 3,
};

union libenv_value {
    uint8_t uint8;
    int32_t int32;
    float float32;
};

struct libenv_tensortype {
    char name[LIBENV_MAX_NAME_LEN];
    enum libenv_scalar_type scalar_type;
    enum libenv_dtype dtype;
    int shape[LIBENV_MAX_NDIM];
    int ndim;
    union libenv_value low;
    union libenv_value high;
};

struct libenv_option {
    char name[LIBENV_MAX_NAME_LEN];
    enum libenv_dtype dtype;
    int count;
    void *data;
};

struct libenv_options {
    struct libenv_option *items;
    int count;
};

/* ob / info / ac are pointer tables indexed [space_idx * num_envs + env_idx]
 * (reference src/vecgame.cpp:30-40); rew / first are flat per-env arrays. */
struct libenv_buffers {
    void **ob;
    float *rew;
    uint8_t *first;
    void **info;
    void **ac;
};

typedef void libenv_env;

/* reference src/vecgame.cpp:43 */
LIBENV_API int libenv_version(void);
/* reference src/vecgame.cpp:47 -- options memory is only valid during the call */
LIBENV_API libenv_env *libenv_make(int num_envs, const struct libenv_options options);
/* reference src/vecgame.cpp:52 -- returns count; fills out_types when non-NULL */
LIBENV_API int libenv_get_tensortypes(libenv_env *handle, enum libenv_space_name name,
                                      struct libenv_tensortype *out_types);
/* reference src/vecgame.cpp:74 -- called once; triggers initial reset + first frame */
LIBENV_API void libenv_set_buffers(libenv_env *handle, struct libenv_buffers *bufs);
/* reference src/vecgame.cpp:85 -- join: on return ob/rew/first/info are valid */
LIBENV_API void libenv_observe(libenv_env *handle);
/* reference src/vecgame.cpp:90 -- reads *ac[e] during the call, steps asynchronously */
LIBENV_API void libenv_act(libenv_env *handle);
/* reference src/vecgame.cpp:95 */
LIBENV_API void libenv_close(libenv_env *handle);

#ifdef __cplusplus
}
#endif
