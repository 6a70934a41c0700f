// EmuPlusPlus.cpp -- see EmuPlusPlus.h (reference: src/EmuPlusPlus.cpp:57-235)
#include "EmuPlusPlus.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>

void emulator::init(const std::string &path, bool pca)
{
	outputPCAValues = pca;
	StateFilePath = path;
	FILE *fptr = fopen(path.c_str(), "r");
	if (!fptr) {
		std::cerr << "error opening statefile: " << path << std::endl;
		gpemu_host_exit(EXIT_FAILURE);
	}
	the_model = load_multi_modelstruct(fptr);
	fclose(fptr);
	the_emulator = alloc_multi_emulator(the_model);
	number_outputs = pca ? the_model->nr : the_model->nt;
	number_params = the_model->nparams;
}

emulator::emulator(std::string path) { init(path, false); }
emulator::emulator(std::string path, bool PcaOnly) { init(path, PcaOnly); }
emulator::~emulator() { free_multi_emulator(the_emulator); }

void emulator::QueryEmulator(const std::vector<std::vector<double> > &xpoints, std::vector<std::vector<double> > &Means,
                             std::vector<std::vector<double> > &Errors)
{
	const size_t np = xpoints.size();
	std::vector<double> flat(np * number_params), m(np * number_outputs), v(np * number_outputs);
	for (size_t q = 0; q < np; q++) {
		if ((int)xpoints[q].size() != number_params) {
			std::cerr << "Error::QueryEmulator called with incorrect number of dimensions in xpoint" << std::endl;
			gpemu_host_exit(EXIT_FAILURE);
		}
		for (int k = 0; k < number_params; k++) flat[q * number_params + k] = xpoints[q][k];
	}
	gsl_matrix view;
	view.size1 = np; view.size2 = number_params; view.tda = number_params; view.data = flat.data(); view.block = NULL; view.owner = 0;
	emulate_points_multi(the_emulator, &view, outputPCAValues ? 1 : 0, m.data(), v.data());
	Means.assign(np, std::vector<double>(number_outputs));
	Errors.assign(np, std::vector<double>(number_outputs));
	for (size_t q = 0; q < np; q++)
		for (int i = 0; i < number_outputs; i++) {
			Means[q][i] = m[q * number_outputs + i];
			Errors[q][i] = sqrt(v[q * number_outputs + i]);      // the reference returns the square root of the variance
		}
}

void emulator::QueryEmulator(const std::vector<double> &xpoint, std::vector<double> &Means, std::vector<double> &Errors)
{
	if ((int)xpoint.size() != number_params) {
		std::cerr << "Error::QueryEmulator called with incorrect number of dimensions in xpoint" << std::endl;
		std::cerr << "xpoint.length: " << xpoint.size() << " emulator->number_params: " << number_params << std::endl;
		gpemu_host_exit(EXIT_FAILURE);
	}
	if (!Means.empty()) { std::cerr << "Error::QueryEmulator called with nonempty Means vector" << std::endl; gpemu_host_exit(EXIT_FAILURE); }
	if (!Errors.empty()) { std::cerr << "Error::QueryEmulator called with nonempty Errors vector" << std::endl; gpemu_host_exit(EXIT_FAILURE); }
	std::vector<std::vector<double> > xs(1, xpoint), mm, ee;
	QueryEmulator(xs, mm, ee);
	Means = mm[0];
	Errors = ee[0];
}

void emulator::getEmulatorPCA(std::vector<double> *pca_evals, std::vector<std::vector<double> > *pca_evecs,
                              std::vector<double> *pca_mean)
{
	const int nt = the_model->nt, nr = the_model->nr;
	pca_evals->assign(nr, 0.0);
	pca_mean->assign(nt, 0.0);
	pca_evecs->assign(nt, std::vector<double>(nr, 0.0));
	for (int i = 0; i < nr; i++) (*pca_evals)[i] = gsl_vector_get(the_model->pca_evals_r, i);
	for (int i = 0; i < nt; i++) {
		(*pca_mean)[i] = gsl_vector_get(the_model->training_mean, i);
		for (int j = 0; j < nr; j++) (*pca_evecs)[i][j] = gsl_matrix_get(the_model->pca_evecs_r, i, j);
	}
}
