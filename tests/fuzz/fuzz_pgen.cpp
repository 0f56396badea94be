// Mutation fuzzer for the host-side .pgen parser / normaliser (plinking_duck_amd/csrc/pgen_file.cpp),
// built with -fsanitize=address,undefined by tests/test_fuzz_host.py.  Every mutated file must be
// either decoded or rejected with an error -- never read or write out of bounds.
//
//   fuzz_pgen <seed.pgen> <scratch path> <iterations> <rng seed>
#include "pgen_file.hpp"

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <random>
#include <vector>

int main(int argc, char **argv) {
	if (argc < 5) {
		return 2;
	}
	std::ifstream in(argv[1], std::ios::binary);
	std::vector<char> seed((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
	if (seed.empty()) {
		return 2;
	}
	const int iterations = std::atoi(argv[3]);
	std::mt19937_64 rng(std::strtoull(argv[4], nullptr, 10));
	int decoded = 0, rejected = 0;
	for (int it = 0; it < iterations; it++) {
		std::vector<char> blob = seed;
		const int edits = 1 + static_cast<int>(rng() % 4);
		for (int e = 0; e < edits; e++) {
			const size_t at = rng() % blob.size();
			switch (rng() % 5) {
			case 0:
				blob[at] = static_cast<char>(rng());
				break;
			case 1:
				blob[at] = static_cast<char>(0xff);
				break;
			case 2:
				blob[at] ^= static_cast<char>(1u << (rng() % 8));
				break;
			case 3:
				blob.resize(at + 1); // truncate
				break;
			default:
				blob[at] = 0;
				break;
			}
		}
		{
			std::ofstream out(argv[2], std::ios::binary | std::ios::trunc);
			out.write(blob.data(), static_cast<std::streamsize>(blob.size()));
		}
		pgh::PgenIndex ix;
		std::string err;
		if (!pgh::ParsePgenIndex(argv[2], "", ix, err)) {
			rejected++;
			continue;
		}
		pgh::RecordFile file;
		if (!file.Open(argv[2], err)) {
			rejected++;
			continue;
		}
		if (ix.RecordBytes() > (1u << 16)) {
			rejected++; // a sample count the product would fail to allocate rows for; nothing to decode here
			continue;
		}
		pgh::Normalizer norm(ix, file);
		const size_t pitch = (ix.RecordBytes() + 15) / 16 * 16;
		const uint32_t rows = ix.variant_ct < 512 ? ix.variant_ct : 512; // a corrupt count must not ask for terabytes
		std::vector<uint8_t> dst(static_cast<size_t>(rows) * pitch + 16);
		bool ok = norm.ExpandRange(0, rows, dst.data(), pitch, err);
		std::vector<uint8_t> row;
		std::vector<uint16_t> dosage;
		std::vector<uint8_t> pp, pi;
		for (uint32_t v = 0; v < rows && v < 8; v++) {
			(void)norm.DecodeDosage(v, row, dosage, err);
			(void)norm.DecodePhase(v, row, pp, pi, err);
		}
		(ok ? decoded : rejected)++;
	}
	std::printf("decoded %d rejected %d\n", decoded, rejected);
	return 0;
}
