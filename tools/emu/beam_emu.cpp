// The beam-search kernel of libdsmi.so (danspeech_amd/csrc/beam_kernel.inc) compiled for the CPU on the SIMT emulation of
// simt.h: the same source, one host thread per GPU thread.  Reads a problem from a binary file written by
// tools/emu/run_beam_emu.py, writes the beams; that script compares them with oracle/beam.py.  A debugging tool (a hang
// or a wrong phase order shows up here, under gdb, instead of on a GPU box); not part of the product, not a CPU path.
//   g++ -O1 -g -std=c++20 -pthread -I danspeech_amd/csrc tools/emu/beam_emu.cpp -o tools/emu/beam_emu
#include "simt.h"
#include <cstdio>
#include <string>
#include <fstream>
#include "lm.h"
#include "lm.cpp.inc"
#include "lm_klm.cpp.inc"
using namespace dsmi;
namespace { alignas(16) unsigned char smem_raw[160 * 1024]; }      // (the kernel declares it inside its anonymous namespace)
#define DSMI_WAIT_STORES() __atomic_thread_fence(__ATOMIC_SEQ_CST)
#include "beam_kernel.inc"

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: beam_emu problem.bin out.bin [threads]\n"); return 2; }
    std::ifstream f(argv[1], std::ios::binary);
    int32_t hd[8];
    f.read((char*)hd, sizeof(hd));
    const int B = hd[0], T = hd[1], C = hd[2], beam = hd[3], blank = hd[4], top_n = hd[5], has_lm = hd[6], nlab = hd[7];
    double dd[3];
    f.read((char*)dd, sizeof(dd));                       // cutoff_prob, alpha, beta
    std::vector<float> probs((size_t)B * T * C);
    f.read((char*)probs.data(), probs.size() * 4);
    std::vector<int32_t> sizes(B);
    f.read((char*)sizes.data(), B * 4);
    std::vector<std::string> labels;
    for (int i = 0; i < nlab; ++i) { int32_t n; f.read((char*)&n, 4); std::string s(n, 0); f.read(&s[0], n); labels.push_back(s); }
    std::string lm_path;
    { int32_t n; f.read((char*)&n, 4); lm_path.resize(n); f.read(&lm_path[0], n); }
    const int BT = argc > 3 ? std::atoi(argv[3]) : 1024;
    HostLM lm;
    if (has_lm) { const std::string msg = lm.load(lm_path, labels); if (!msg.empty()) { std::fprintf(stderr, "%s\n", msg.c_str()); return 3; } }
    int space = -2;
    for (int i = 0; i < nlab; ++i) if (labels[i] == " ") space = i;
    const int ncap = 2 + T * beam;
    std::vector<NodeRec> nodes((size_t)B * ncap);
    std::vector<int32_t> tok((size_t)B * beam * T), step((size_t)B * beam * T), len((size_t)B * beam), nout(B), dbg((size_t)B * 4);
    std::vector<double> score((size_t)B * beam);
    BeamArgs a{};
    a.probs = probs.data(); a.sizes = sizes.data(); a.T = T; a.C = C; a.blank = blank; a.space = space; a.beam = beam;
    a.cutoff_top_n = top_n; a.cutoff_prob = (float)dd[0]; a.has_lm = has_lm; a.order = has_lm ? lm.order : 1; a.alpha = dd[1]; a.beta = dd[2];
    a.lm = lm.view(); a.trie_next = lm.trie_next.data(); a.trie_word = lm.trie_word.data(); a.unk = lm.unk; a.bos = lm.bos;
    a.ncap = ncap; a.nodes = nodes.data(); a.dbg = dbg.data();
    a.out_tok = tok.data(); a.out_step = step.data(); a.out_len = len.data(); a.out_n = nout.data(); a.out_score = score.data();
    const size_t NMAX = (size_t)beam * (C + 1);
    if (carve(beam, C, BT).bytes > sizeof(smem_raw)) { std::fprintf(stderr, "does not fit\n"); return 4; }
    const int EW = (beam + 63) / 64, RW = has_lm ? 2 * EW : EW;
    if (BT <= 64 * RW) { std::fprintf(stderr, "too few threads for this beam\n"); return 4; }
    const size_t per = ((size_t)beam * C + (BT - 64 * RW) - 1) / (BT - 64 * RW);       // pairs per pair thread
    if (BT == 1024) {
        if (per <= 3) simt::launch(B, BT, [&]() { beam_kernel<1024, 3>(a); });
        else simt::launch(B, BT, [&]() { beam_kernel<1024, 6>(a); });
    } else if (BT == 512) {
        if (per <= 6) simt::launch(B, BT, [&]() { beam_kernel<512, 6>(a); });
        else simt::launch(B, BT, [&]() { beam_kernel<512, 11>(a); });
    } else {
        if (per > 9) { std::fprintf(stderr, "too many pairs per thread for the 192-thread build\n"); return 4; }
        simt::launch(B, 192, [&]() { beam_kernel<192, 9>(a); });      // three waves: entries, (entry, space) pairs, the other pairs
    }
    std::ofstream o(argv[2], std::ios::binary);
    o.write((char*)tok.data(), tok.size() * 4); o.write((char*)step.data(), step.size() * 4); o.write((char*)len.data(), len.size() * 4);
    o.write((char*)nout.data(), nout.size() * 4); o.write((char*)score.data(), score.size() * 8); o.write((char*)dbg.data(), dbg.size() * 4);
    std::fprintf(stderr, "emulated %d utterances on %d threads each: revivals %d, walk hops %d, list rankings %d, full rankings %d\n", B, BT,
                 dbg[0], dbg[1], dbg[2], dbg[3]);
    return 0;
}
