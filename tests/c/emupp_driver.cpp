// drives the EmuPlusPlus-compatible C++ class: emupp_driver SNAPSHOT QUERY_FILE [pca]
#include "EmuPlusPlus.h"
#include <cstdio>
#include <cstdlib>
#include <fstream>
int main(int argc, char **argv)
{
	if (argc < 3) return 2;
	const bool pca = argc > 3;
	emulator emu(argv[1], pca);
	std::ifstream in(argv[2]);
	std::vector<std::vector<double> > pts;
	std::vector<double> p(emu.number_params);
	for (;;) {
		int k = 0;
		for (; k < emu.number_params && (in >> p[k]); k++) {}
		if (k < emu.number_params) break;
		pts.push_back(p);
	}
	printf("info %d %d %d %d\n", emu.number_params, emu.number_outputs, emu.getRegressionOrder(), emu.getCovFnIndex());
	for (size_t q = 0; q < pts.size(); q++) {
		std::vector<double> m, e;
		emu.QueryEmulator(pts[q], m, e);
		printf("single");
		for (size_t i = 0; i < m.size(); i++) printf(" %.17g %.17g", m[i], e[i]);
		printf("\n");
	}
	std::vector<std::vector<double> > mm, ee;
	emu.QueryEmulator(pts, mm, ee);
	for (size_t q = 0; q < pts.size(); q++) {
		printf("batch");
		for (size_t i = 0; i < mm[q].size(); i++) printf(" %.17g %.17g", mm[q][i], ee[q][i]);
		printf("\n");
	}
	std::vector<double> ev, mean;
	std::vector<std::vector<double> > evec;
	emu.getEmulatorPCA(&ev, &evec, &mean);
	printf("pca %zu %zu %zu\n", ev.size(), evec.size(), mean.size());
	return 0;
}
