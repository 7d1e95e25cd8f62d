// Test harness around the REFERENCE's own FASTR scoring functions.  TEST INFRASTRUCTURE ONLY.
//
// It includes apps/fastr/IndelRealigner.cpp from the reference tree where it lies (the
// compiler finds it through -I; nothing is copied) because the functions under test and the
// lookup table they use are file-static there:
//   mismatchQualitySumIgnoreCigar   apps/fastr/IndelRealigner.cpp:264-295
//   findBestOffset                  apps/fastr/IndelRealigner.cpp:502-542
//   ch_lookup initialisation        apps/fastr/IndelRealigner.cpp:1142-1149
//   SWPairwiseAlignment             apps/fastr/SWPairwiseAlignment.cpp:64-71,94-399 (its own translation unit)
// and exposes them through a C entry point.  Built by oracle/build_ref.sh into
// oracle/_ref/libfastr_ref.so.
#define main fastr_reference_main
#include "IndelRealigner.cpp"
#undef main

#include <string>

extern "C" int fastr_ref_find_best_offset(const char *ref, int ref_len, const char *bases, const char *phred,
                                          int read_len, int original_alignment, int *best_index, int *best_score)
{
    static bool ready = false;
    if (!ready) {
        for (short i = 0; i < 128; i++) ch_lookup[i] = isRegularBase((char)i) ? 1 : 0;
        ready = true;
    }
    BamAlignment r;
    r.Length = read_len;
    r.Position = original_alignment;
    r.AlignedBases.assign(bases, read_len);
    r.QueryBases.assign(bases, read_len);
    std::string q(read_len, '!');
    for (int i = 0; i < read_len; ++i) q[i] = (char)(phred[i] + 33);
    r.Qualities = q;
    r.CigarData.push_back(CigarOp('M', read_len));
    AlignedRead ar(&r);
    std::pair<int, int> p = findBestOffset((const byte *)ref, ref_len, ar, 0);
    *best_index = p.first;
    *best_score = p.second;
    return 0;
}

// The reference's pairwise aligner as createAndAddAlternateConsensus calls it
// (apps/fastr/IndelRealigner.cpp:366-371): sequence 1 = reference window, sequence 2 = read.
extern "C" int fastr_ref_sw_align(const char *ref, int ref_len, const char *read, int read_len, double match,
                                  double mismatch, double open, double extend, int *alignment_offset,
                                  char *op_type, int *op_len, int max_ops)
{
    SWPairwiseAlignment sw((const byte *)ref, (unsigned)ref_len, (const byte *)read, (unsigned)read_len, match,
                           mismatch, open, extend);
    *alignment_offset = sw.getAlignmentStart2wrt1();
    Cigar c = sw.getCigar();
    for (size_t k = 0; k < c.size() && (int)k < max_ops; ++k) {
        op_type[k] = c[k].Type;
        op_len[k] = (int)c[k].Length;
    }
    return (int)c.size();
}
