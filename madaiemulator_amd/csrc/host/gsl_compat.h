/*
 * gsl_compat.h -- the handful of GSL container types the reference's libEmu
 * interface is written in (gsl_vector, gsl_matrix, views, gsl_rng), with the
 * SAME memory layout as GSL 1.x/2.x so that modelstruct / optstruct /
 * emulator_struct keep the reference's field layout (modelstruct.h:28-98,
 * optstruct.h:25-88, emulator_struct.h:20-29).  No GSL numerics here: all
 * dense linear algebra runs on the GPU behind include/gpemu.h.
 *
 * A site that has the real GSL compiles the host layer with
 * -DGPEMU_USE_SYSTEM_GSL and links -lgsl instead.
 */
#ifndef GPEMU_GSL_COMPAT_H
#define GPEMU_GSL_COMPAT_H

#ifdef GPEMU_USE_SYSTEM_GSL
#include <gsl/gsl_vector.h>
#include <gsl/gsl_matrix.h>
#include <gsl/gsl_rng.h>
#include <gsl/gsl_math.h>
#include <gsl/gsl_errno.h>
#else

#include <stddef.h>
#include <math.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { size_t size; double *data; } gsl_block;

typedef struct {
	size_t size;
	size_t stride;
	double *data;
	gsl_block *block;
	int owner;
} gsl_vector;

typedef struct {
	size_t size1;
	size_t size2;
	size_t tda;
	double *data;
	gsl_block *block;
	int owner;
} gsl_matrix;

typedef struct { gsl_vector vector; } gsl_vector_view;
typedef struct { gsl_matrix matrix; } gsl_matrix_view;

#define GSL_NAN (NAN)
#define GSL_SUCCESS  0
#define GSL_CONTINUE (-2)
#define GSL_EDOM     1
#define GSL_ENOPROG  27

gsl_vector *gsl_vector_alloc(size_t n);
gsl_vector *gsl_vector_calloc(size_t n);
void gsl_vector_free(gsl_vector *v);
void gsl_vector_set_zero(gsl_vector *v);
int gsl_vector_memcpy(gsl_vector *dst, const gsl_vector *src);

gsl_matrix *gsl_matrix_alloc(size_t n1, size_t n2);
gsl_matrix *gsl_matrix_calloc(size_t n1, size_t n2);
void gsl_matrix_free(gsl_matrix *m);
void gsl_matrix_set_zero(gsl_matrix *m);
int gsl_matrix_memcpy(gsl_matrix *dst, const gsl_matrix *src);
gsl_vector_view gsl_matrix_row(gsl_matrix *m, size_t i);
gsl_vector_view gsl_matrix_column(gsl_matrix *m, size_t j);

static inline double gsl_vector_get(const gsl_vector *v, size_t i) { return v->data[i * v->stride]; }
static inline void gsl_vector_set(gsl_vector *v, size_t i, double x) { v->data[i * v->stride] = x; }
static inline double *gsl_vector_ptr(gsl_vector *v, size_t i) { return v->data + i * v->stride; }
static inline double gsl_matrix_get(const gsl_matrix *m, size_t i, size_t j) { return m->data[i * m->tda + j]; }
static inline void gsl_matrix_set(gsl_matrix *m, size_t i, size_t j, double x) { m->data[i * m->tda + j] = x; }
static inline double *gsl_matrix_ptr(gsl_matrix *m, size_t i, size_t j) { return m->data + i * m->tda + j; }

/* gsl_rng_default = mt19937; gsl_rng_uniform = genrand_int32()/2^32 in [0,1) */
typedef struct { unsigned long mt[624]; int mti; } gsl_rng;
typedef struct { const char *name; } gsl_rng_type;
extern const gsl_rng_type *gsl_rng_default;
gsl_rng *gsl_rng_alloc(const gsl_rng_type *T);
void gsl_rng_set(gsl_rng *r, unsigned long seed);
unsigned long gsl_rng_get(gsl_rng *r);
double gsl_rng_uniform(gsl_rng *r);
void gsl_rng_free(gsl_rng *r);

#ifdef __cplusplus
}
#endif
#endif /* GPEMU_USE_SYSTEM_GSL */
#endif
