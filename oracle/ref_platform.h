/* ref_platform.h -- force-included when compiling the reference's kcf.cpp /
 * drawlib.c on Linux.  The reference targets MSVC, whose CRT provides
 * _aligned_malloc/_aligned_free; its own top/cnntype.h:14-16 carries the
 * (commented-out) Linux mapping.  These two defines are that mapping; no
 * reference source is modified or copied. */
#include <stdlib.h>
#define _aligned_malloc(sz, al) aligned_alloc((al), ((((size_t)(sz)) + (al) - 1) / (al)) * (al))
#define _aligned_free(p) free(p)
