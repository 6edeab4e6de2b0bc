// CLIP byte-pair-encoding tokenizer -- the host-side twin of src/tokenizer.rs (SimpleTokenizer).
// the step before the hot path (prompt -> token ids -> CLIP -> context).
#pragma once

#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

namespace sdmi {

class Tokenizer {
public:
    // SimpleTokenizer::new (tokenizer.rs:85-120); the reference opens "bpe_simple_vocab_16e6.txt" in the
    // working directory, here the path is an argument.  Throws sdmi::Error (SDMI_ERR_IO) on a missing file.
    explicit Tokenizer(const std::string& merges_path);

    std::vector<int32_t> encode(const std::string& text_utf8) const;   // tokenizer.rs:168-189
    std::string decode(const int32_t* ids, size_t n) const;            // tokenizer.rs:191-196
    int vocab_size() const { return (int)decoder_.size(); }
    int start_token() const { return sot_; }   // <|startoftext|>
    int end_token() const { return eot_; }     // <|endoftext|>

private:
    std::vector<std::string> bpe(const std::vector<std::string>& symbols) const;  // tokenizer.rs:122-166

    std::string byte_to_sym_[256];                          // bytes_to_unicode (tokenizer.rs:6-28), as UTF-8
    std::unordered_map<uint32_t, uint8_t> sym_to_byte_;     // code point -> byte
    std::unordered_map<std::string, int32_t> encoder_;      // vocab string -> id
    std::vector<std::string> decoder_;                      // id -> vocab string
    std::unordered_map<std::string, int32_t> ranks_;        // "first second" -> merge rank
    int sot_ = -1, eot_ = -1;
};

// Unicode helpers shared with the tests (csrc/tokenizer.cpp)
std::vector<uint32_t> utf8_decode_lossy(const std::string& s);
void utf8_append(std::string& out, uint32_t cp);
std::vector<uint32_t> unicode_lowercase(const std::vector<uint32_t>& cps);  // str::to_lowercase
bool unicode_is_letter(uint32_t cp);
bool unicode_is_number(uint32_t cp);
bool unicode_is_space(uint32_t cp);

}  // namespace sdmi
