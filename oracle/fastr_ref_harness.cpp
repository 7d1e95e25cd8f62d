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

// Regions built from the reference's OWN objects (BamAlignment, AlignedRead, ReadBin, Consensus,
// ConsensusHashTable), flattened exactly as clean_gpu_prep flattens a batch
// (apps/fastr/IndelRealigner.cpp:714-752: getReadBases(), getBaseQualities(), getReadLength(),
// getOriginalAlignmentStart() - leftmostIndex, the consensus strings in toArray() order) and scored pair
// by pair as the per-region CPU path scores them (clean, :857-880: findBestOffset(consensus->str,
// consensus->strLen, *toTest, leftmostIndex)).  Results are ordered [region][consensus][read].
//
// Inputs, all regions back to back: per region the number of reads in the bin and how many of them
// (the LAST n_alt ones) are alternate reads; per read its position, length, leading soft clip, bases
// and phred qualities; per region the candidate consensus strings (duplicates allowed: the hash table
// drops what it considers equal, :225-237 of ConsensusHashTable.cpp).
extern "C" int fastr_ref_regions(int n_regions, const int *n_bin_reads, const int *n_alt_reads, const int *read_pos,
                                 const int *read_len, const int *soft_clip, const char *read_bases,
                                 const char *read_phred, const int *n_candidates, const int *cand_len,
                                 const char *cand_bases, int stride, unsigned char *h_readSeq, unsigned char *h_quals,
                                 int *ReadLength, int *originalAlignment, int *h_n_consensus, int *h_altReads,
                                 int *h_refLens, unsigned char *h_ref, int *totals, int *best_index, int *best_score)
{
    for (short i = 0; i < 128; i++) ch_lookup[i] = isRegularBase((char)i) ? 1 : 0;
    int read_at = 0, base_at = 0, cand_at = 0, cand_base_at = 0;
    int reads_off = 0, ref_off = 0, consensus_off = 0, score_off = 0;
    for (int t = 0; t < n_regions; ++t) {
        ReadBin bin("chr", NULL);
        std::vector<BamAlignment *> owned;
        std::vector<AlignedRead *> altReads;
        for (int j = 0; j < n_bin_reads[t]; ++j, ++read_at) {
            const int len = read_len[read_at], clip = soft_clip[read_at];
            BamAlignment *r = new BamAlignment();
            r->Length = len;
            r->Position = read_pos[read_at];
            r->AlignedBases.assign(read_bases + base_at + clip, len - clip);  // bamtools strips soft clips here
            r->QueryBases.assign(read_bases + base_at, len);
            std::string q(len, '!');
            for (int i = 0; i < len; ++i) q[i] = (char)(read_phred[base_at + i] + 33);
            r->Qualities = q;
            if (clip > 0) r->CigarData.push_back(CigarOp('S', clip));
            r->CigarData.push_back(CigarOp('M', len - clip));
            base_at += len;
            owned.push_back(r);
            bin.add(r);
            if (j >= n_bin_reads[t] - n_alt_reads[t]) altReads.push_back(new AlignedRead(r));
        }
        ConsensusHashTable *altConsensus = new ConsensusHashTable();
        for (int k = 0; k < n_candidates[t]; ++k, ++cand_at) {
            Cigar c;
            c.push_back(CigarOp('M', cand_len[cand_at]));
            byte *own = new byte[cand_len[cand_at]];   // a Consensus deletes its string (Consensus.cpp:78-81)
            memcpy(own, cand_bases + cand_base_at, cand_len[cand_at]);
            Consensus *cons = new Consensus(own, cand_len[cand_at], c, 0);
            cand_base_at += cand_len[cand_at];
            if (!altConsensus->put(cons)) delete cons;
        }
        // ---- the flattening loop of clean_gpu_prep, for this region
        std::vector<Consensus *> finalConsensus;
        altConsensus->toArray(finalConsensus);
        const int leftmostIndex = bin.getStart();
        h_altReads[t] = (int)altReads.size();
        for (unsigned j = 0; j < altReads.size(); ++j) {
            AlignedRead *toTest = altReads[j];
            byte *readSeq = toTest->getReadBases();
            byte *quals = toTest->getBaseQualities();
            ReadLength[reads_off] = toTest->getReadLength();
            originalAlignment[reads_off] = toTest->getOriginalAlignmentStart() - leftmostIndex;
            if (ReadLength[reads_off] > stride) return -1;
            memcpy(h_readSeq + (size_t)reads_off * stride, readSeq, ReadLength[reads_off]);
            memcpy(h_quals + (size_t)reads_off * stride, quals, ReadLength[reads_off]);
            reads_off += 1;
        }
        h_n_consensus[t] = (int)finalConsensus.size();
        for (size_t k = 0; k < finalConsensus.size(); ++k) {
            Consensus *consensus = finalConsensus[k];
            memcpy(h_ref + ref_off, consensus->str, consensus->strLen);
            ref_off += consensus->strLen;
            h_refLens[consensus_off++] = consensus->strLen;
            // ---- the per-region CPU path's scoring of this consensus
            for (unsigned j = 0; j < altReads.size(); ++j) {
                std::pair<int, int> p = findBestOffset(consensus->str, consensus->strLen, *altReads[j], leftmostIndex);
                best_index[score_off] = p.first;
                best_score[score_off] = p.second;
                score_off += 1;
            }
        }
        for (size_t j = 0; j < altReads.size(); ++j) delete altReads[j];
        delete altConsensus;   // the table's entries delete their consensuses (ConsensusHashTable.cpp:57-60)
        for (size_t j = 0; j < owned.size(); ++j) delete owned[j];
    }
    totals[0] = reads_off;
    totals[1] = ref_off;
    totals[2] = score_off;
    totals[3] = consensus_off;
    return 0;
}
