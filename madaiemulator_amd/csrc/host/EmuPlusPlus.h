// EmuPlusPlus.h -- C++ query class over a trained emulator, same public interface as the reference's
// src/EmuPlusPlus.h:31-65 (class emulator: constructors from a MODEL_SNAPSHOT_FILE, QueryEmulator returning
// means and sqrt(variance), getEmulatorPCA), plus a batched QueryEmulator: MCMC drivers that hold many
// proposals should pass them in one call -- the posterior sweep is one GEMM over the query block.
#ifndef GPEMU_EMUPLUSPLUS_H
#define GPEMU_EMUPLUSPLUS_H

#include <string>
#include <vector>
extern "C" {
#include "libemu.h"
}

class emulator {
public:
	explicit emulator(std::string StateFilePath);                 // outputs in the observable (Y) space
	emulator(std::string StateFilePath, bool PcaOnly);            // PcaOnly: outputs left in the PCA space
	~emulator();
	// Means / Errors must be empty on entry; Errors = sqrt(variance)  (EmuPlusPlus.cpp:137-178)
	void QueryEmulator(const std::vector<double> &xpoint, std::vector<double> &Means, std::vector<double> &Errors);
	// batch: one row per query point; Means[q], Errors[q] have number_outputs entries
	void QueryEmulator(const std::vector<std::vector<double> > &xpoints, std::vector<std::vector<double> > &Means,
	                   std::vector<std::vector<double> > &Errors);
	void getEmulatorPCA(std::vector<double> *pca_evals, std::vector<std::vector<double> > *pca_evecs,
	                    std::vector<double> *pca_mean);
	int getRegressionOrder(void) { return the_model->regression_order; }
	int getCovFnIndex(void) { return the_model->cov_fn_index; }
	int number_params;
	int number_outputs;

private:
	void init(const std::string &path, bool pca);
	bool outputPCAValues;
	std::string StateFilePath;
	multi_modelstruct *the_model;
	multi_emulator *the_emulator;
};
#endif
