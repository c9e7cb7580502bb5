// Result-side text assembly of the timestamp path — behaviourally `funasr::PostProcess` (onnxruntime/src/util.cpp:720-836):
// hypothesis characters + their time stamps -> "<text> | <b0>, <e0>,<b1>, <e1>..." (SURVEY §8 row f4).  Special tokens are
// dropped, BPE pieces ending in "@@" are glued to the following piece (a word's stamp runs from its first piece's begin to
// its last piece's end), Latin words are separated by single spaces, CJK characters are not.  Host string handling only.
#pragma once
#include <string>
#include <vector>

namespace pfhip_host {

// raw_char[i] is the vocabulary string of token i, stamps[i] = {begin_s, end_s} of the same token (same length).
std::string PostProcess(const std::vector<std::string>& raw_char, const std::vector<std::vector<float>>& stamps);

}  // namespace pfhip_host
