// Harness for the WFST hand-off of ParaformerHip::Forward (paraformer.cpp:563-579, paraformer-torch.cpp:431-466):
//   decoder_handoff <model_dir> <pcm_s16_file> <n_utts> <rows_out.bin> <with_lm 0|1> <input_finished 0|1>
// The PCM file is cut into n_utts utterances (the last one 300 samples: no feature frame), Forward runs ONCE over the batch
// with a recording decoder as FUNASR_DEC_HANDLE.  Every call the adapter makes on the decoder is logged to stdout in order
// ("search k len V", "finalize k is_stamp n_alphas n_peaks", "start"), the rows handed to Search are appended to
// rows_out.bin (int32 len, int32 V, len*V floats per call), then "result i <text>" per utterance and "ids i ..." (greedy).
// with_lm = 1 calls InitLm first (stand-alone build: the LM file only has to exist; the decoder owns the graph).
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "paraformer_hip.h"

namespace {
struct RecordingDecoder : funasr::Decoder {
  std::FILE* rows;
  int searches = 0, finals = 0;
  explicit RecordingDecoder(std::FILE* f) : rows(f) {}
  void StartUtterance() override { std::printf("start\n"); }
  std::string Search(float* in, int len, int64_t token_nums) override {
    std::printf("search %d %d %lld\n", searches, len, (long long)token_nums);
    const int32_t hdr[2] = {len, (int32_t)token_nums};
    std::fwrite(hdr, 4, 2, rows);
    std::fwrite(in, 4, (size_t)len * token_nums, rows);
    return "S" + std::to_string(searches++);
  }
  std::string FinalizeDecode(bool is_stamp, std::vector<float> us_alphas, std::vector<float> us_cif_peak) override {
    std::printf("finalize %d %d %zu %zu\n", finals, is_stamp ? 1 : 0, us_alphas.size(), us_cif_peak.size());
    return "F" + std::to_string(finals++);
  }
};
}  // namespace

int main(int argc, char** argv) {
  if (argc < 7) { std::fprintf(stderr, "usage: %s model_dir pcm_s16 n_utts rows.bin with_lm input_finished\n", argv[0]); return 2; }
  const std::string dir = argv[1];
  const int n_utts = std::atoi(argv[3]);
  const bool with_lm = std::atoi(argv[5]) != 0, fin = std::atoi(argv[6]) != 0;
  std::ifstream f(argv[2], std::ios::binary);
  std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  const size_t total = raw.size() / 2;
  std::vector<float> pcm(total);
  for (size_t i = 0; i < total; ++i) pcm[i] = (float)reinterpret_cast<const int16_t*>(raw.data())[i] / 32768.0f;      // audio.cpp:797-805
  // utterance k of the first n_utts - 1: samples [k*step, (k+1)*step - 1000*k); the last: 300 samples
  std::vector<float*> din(n_utts);
  std::vector<int> len(n_utts);
  const size_t step = n_utts > 1 ? (total - 300) / (n_utts - 1) : total;
  for (int k = 0; k < n_utts; ++k) {
    din[k] = pcm.data() + (size_t)k * step;
    len[k] = k + 1 < n_utts || n_utts == 1 ? (int)step - 1000 * k : 300;
  }
  funasr::ParaformerHip model;
  const std::string tokens = dir + "/tokens.json";
  model.InitAsr(dir + "/model.pfhip.bin", "", dir + "/model.pfhip.json", std::ifstream(tokens) ? tokens : std::string(), 1);
  funasr::ParaformerHipBase* asr_handle = &model;
  if (with_lm) asr_handle->InitLm(argv[2], "", "");          // offline-stream.cpp:102 (three arguments)
  std::FILE* rows = std::fopen(argv[4], "wb");
  if (!rows) return 1;
  RecordingDecoder dec(rows);
  const std::vector<std::vector<float>> no_hw;
  const std::vector<std::string> res = asr_handle->Forward(din.data(), len.data(), fin, no_hw, static_cast<funasr::Decoder*>(&dec), n_utts);
  std::fclose(rows);
  for (size_t i = 0; i < res.size(); ++i) std::printf("result %zu %s\n", i, res[i].c_str());
  const auto& ids = model.LastTokenIds();
  for (size_t i = 0; i < ids.size(); ++i) {
    std::printf("ids %zu", i);
    for (int v : ids[i]) std::printf(" %d", v);
    std::printf("\n");
  }
  return 0;
}
