/* containers.c -- GSL-layout vector/matrix containers and the mt19937 generator (gsl_compat.h). */
#include "gsl_compat.h"

#ifndef GPEMU_USE_SYSTEM_GSL
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

extern void gpemu_host_exit(int status) __attribute__((noreturn));   /* fatal.c */

static gsl_block *block_alloc(size_t n, int zero)
{
	gsl_block *b = (gsl_block *)malloc(sizeof(gsl_block));
	if (!b) { fprintf(stderr, "gsl_compat: out of memory\n"); gpemu_host_exit(EXIT_FAILURE); }
	b->size = n;
	b->data = (double *)(zero ? calloc(n ? n : 1, sizeof(double)) : malloc((n ? n : 1) * sizeof(double)));
	if (!b->data) { fprintf(stderr, "gsl_compat: out of memory\n"); gpemu_host_exit(EXIT_FAILURE); }
	return b;
}

static gsl_vector *vector_new(size_t n, int zero)
{
	gsl_vector *v = (gsl_vector *)malloc(sizeof(gsl_vector));
	v->block = block_alloc(n, zero);
	v->data = v->block->data;
	v->size = n;
	v->stride = 1;
	v->owner = 1;
	return v;
}

gsl_vector *gsl_vector_alloc(size_t n) { return vector_new(n, 0); }
gsl_vector *gsl_vector_calloc(size_t n) { return vector_new(n, 1); }

void gsl_vector_free(gsl_vector *v)
{
	if (!v) return;
	if (v->owner && v->block) { free(v->block->data); free(v->block); }
	free(v);
}

void gsl_vector_set_zero(gsl_vector *v)
{
	for (size_t i = 0; i < v->size; i++) v->data[i * v->stride] = 0.0;
}

int gsl_vector_memcpy(gsl_vector *dst, const gsl_vector *src)
{
	if (dst->size != src->size) return GSL_EDOM;
	for (size_t i = 0; i < src->size; i++) dst->data[i * dst->stride] = src->data[i * src->stride];
	return GSL_SUCCESS;
}

static gsl_matrix *matrix_new(size_t n1, size_t n2, int zero)
{
	gsl_matrix *m = (gsl_matrix *)malloc(sizeof(gsl_matrix));
	m->block = block_alloc(n1 * n2, zero);
	m->data = m->block->data;
	m->size1 = n1;
	m->size2 = n2;
	m->tda = n2;
	m->owner = 1;
	return m;
}

gsl_matrix *gsl_matrix_alloc(size_t n1, size_t n2) { return matrix_new(n1, n2, 0); }
gsl_matrix *gsl_matrix_calloc(size_t n1, size_t n2) { return matrix_new(n1, n2, 1); }

void gsl_matrix_free(gsl_matrix *m)
{
	if (!m) return;
	if (m->owner && m->block) { free(m->block->data); free(m->block); }
	free(m);
}

void gsl_matrix_set_zero(gsl_matrix *m)
{
	for (size_t i = 0; i < m->size1; i++) memset(m->data + i * m->tda, 0, m->size2 * sizeof(double));
}

int gsl_matrix_memcpy(gsl_matrix *dst, const gsl_matrix *src)
{
	if (dst->size1 != src->size1 || dst->size2 != src->size2) return GSL_EDOM;
	for (size_t i = 0; i < src->size1; i++)
		memcpy(dst->data + i * dst->tda, src->data + i * src->tda, src->size2 * sizeof(double));
	return GSL_SUCCESS;
}

gsl_vector_view gsl_matrix_row(gsl_matrix *m, size_t i)
{
	gsl_vector_view v;
	v.vector.size = m->size2;
	v.vector.stride = 1;
	v.vector.data = m->data + i * m->tda;
	v.vector.block = m->block;
	v.vector.owner = 0;
	return v;
}

gsl_vector_view gsl_matrix_column(gsl_matrix *m, size_t j)
{
	gsl_vector_view v;
	v.vector.size = m->size1;
	v.vector.stride = m->tda;
	v.vector.data = m->data + j;
	v.vector.block = m->block;
	v.vector.owner = 0;
	return v;
}

/* mt19937 (Matsumoto & Nishimura 1998), the published reference algorithm; GSL's default seed is 4357 */
static const gsl_rng_type mt_type = {"mt19937"};
const gsl_rng_type *gsl_rng_default = &mt_type;

void gsl_rng_set(gsl_rng *r, unsigned long seed)
{
	if (seed == 0) seed = 4357;
	r->mt[0] = seed & 0xffffffffUL;
	for (int i = 1; i < 624; i++)
		r->mt[i] = (1812433253UL * (r->mt[i - 1] ^ (r->mt[i - 1] >> 30)) + (unsigned long)i) & 0xffffffffUL;
	r->mti = 624;
}

gsl_rng *gsl_rng_alloc(const gsl_rng_type *T)
{
	(void)T;
	gsl_rng *r = (gsl_rng *)malloc(sizeof(gsl_rng));
	gsl_rng_set(r, 0);
	return r;
}

unsigned long gsl_rng_get(gsl_rng *r)
{
	static const unsigned long mag01[2] = {0x0UL, 0x9908b0dfUL};
	unsigned long y;
	if (r->mti >= 624) {
		int kk;
		for (kk = 0; kk < 624 - 397; kk++) {
			y = (r->mt[kk] & 0x80000000UL) | (r->mt[kk + 1] & 0x7fffffffUL);
			r->mt[kk] = r->mt[kk + 397] ^ (y >> 1) ^ mag01[y & 1UL];
		}
		for (; kk < 623; kk++) {
			y = (r->mt[kk] & 0x80000000UL) | (r->mt[kk + 1] & 0x7fffffffUL);
			r->mt[kk] = r->mt[kk + (397 - 624)] ^ (y >> 1) ^ mag01[y & 1UL];
		}
		y = (r->mt[623] & 0x80000000UL) | (r->mt[0] & 0x7fffffffUL);
		r->mt[623] = r->mt[396] ^ (y >> 1) ^ mag01[y & 1UL];
		r->mti = 0;
	}
	y = r->mt[r->mti++];
	y ^= (y >> 11);
	y ^= (y << 7) & 0x9d2c5680UL;
	y ^= (y << 15) & 0xefc60000UL;
	y ^= (y >> 18);
	return y & 0xffffffffUL;
}

double gsl_rng_uniform(gsl_rng *r) { return gsl_rng_get(r) / 4294967296.0; }
void gsl_rng_free(gsl_rng *r) { free(r); }

#endif /* GPEMU_USE_SYSTEM_GSL */
