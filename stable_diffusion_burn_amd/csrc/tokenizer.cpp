// CLIP BPE tokenizer (host code).  Follows src/tokenizer.rs function by function; the Unicode properties
// Rust gets from std / the regex crate come from generated range tables (unicode_tables.inc).
#include "tokenizer.hpp"

#include <algorithm>
#include <climits>
#include <fstream>
#include <sstream>

#include "error.hpp"   // sdmi::Error, status codes

namespace sdmi {

#include "unicode_tables.inc"

// ---- Unicode helpers ---------------------------------------------------------------------------------
template <size_t N>
static bool in_ranges(const uint32_t (&r)[N][2], uint32_t cp) {
    size_t lo = 0, hi = N;
    while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        if (cp < r[mid][0]) hi = mid;
        else if (cp > r[mid][1]) lo = mid + 1;
        else return true;
    }
    return false;
}
bool unicode_is_letter(uint32_t cp) { return in_ranges(kLetterRanges, cp); }
bool unicode_is_number(uint32_t cp) { return in_ranges(kNumberRanges, cp); }
bool unicode_is_space(uint32_t cp) { return in_ranges(kWhiteSpaceRanges, cp); }
static bool is_cased(uint32_t cp) { return in_ranges(kCasedRanges, cp); }
static bool is_case_ignorable(uint32_t cp) { return in_ranges(kCaseIgnorableRanges, cp); }

static uint32_t simple_lower(uint32_t cp) {
    if (cp < 0x80) return (cp >= 'A' && cp <= 'Z') ? cp + 32 : cp;
    constexpr size_t n = sizeof(kLowerMap) / sizeof(kLowerMap[0]);
    size_t lo = 0, hi = n;
    while (lo < hi) {
        const size_t mid = (lo + hi) / 2;
        if (kLowerMap[mid][0] < cp) lo = mid + 1;
        else hi = mid;
    }
    return (lo < n && kLowerMap[lo][0] == cp) ? kLowerMap[lo][1] : cp;
}

// str::to_lowercase: simple mappings, U+0130 -> "i̇", and U+03A3 -> U+03C2 at the end of a word
// (Final_Sigma: preceded by a cased letter and not followed by one, skipping case-ignorable characters).
std::vector<uint32_t> unicode_lowercase(const std::vector<uint32_t>& s) {
    std::vector<uint32_t> out;
    out.reserve(s.size());
    for (size_t i = 0; i < s.size(); ++i) {
        const uint32_t cp = s[i];
        if (cp == 0x130) { out.push_back(0x69); out.push_back(0x307); continue; }
        if (cp == 0x3A3) {
            bool before = false, after = false;
            for (size_t j = i; j-- > 0;) {
                if (is_case_ignorable(s[j])) continue;
                before = is_cased(s[j]);
                break;
            }
            for (size_t j = i + 1; j < s.size(); ++j) {
                if (is_case_ignorable(s[j])) continue;
                after = is_cased(s[j]);
                break;
            }
            out.push_back((before && !after) ? 0x3C2 : 0x3C3);
            continue;
        }
        out.push_back(simple_lower(cp));
    }
    return out;
}

// String::from_utf8_lossy semantics: every maximal invalid prefix becomes U+FFFD
std::vector<uint32_t> utf8_decode_lossy(const std::string& s) {
    std::vector<uint32_t> out;
    const unsigned char* p = reinterpret_cast<const unsigned char*>(s.data());
    const size_t n = s.size();
    size_t i = 0;
    while (i < n) {
        const unsigned char c = p[i];
        if (c < 0x80) { out.push_back(c); ++i; continue; }
        int len = 0;
        uint32_t cp = 0, min = 0;
        unsigned char lo2 = 0x80, hi2 = 0xBF;   // allowed range of the second byte
        if (c >= 0xC2 && c <= 0xDF) { len = 2; cp = c & 0x1F; min = 0x80; }
        else if (c >= 0xE0 && c <= 0xEF) { len = 3; cp = c & 0x0F; min = 0x800; if (c == 0xE0) lo2 = 0xA0; if (c == 0xED) hi2 = 0x9F; }
        else if (c >= 0xF0 && c <= 0xF4) { len = 4; cp = c & 0x07; min = 0x10000; if (c == 0xF0) lo2 = 0x90; if (c == 0xF4) hi2 = 0x8F; }
        else { out.push_back(0xFFFD); ++i; continue; }
        size_t k = 1;
        bool ok = true;
        for (; k < (size_t)len; ++k) {
            if (i + k >= n) { ok = false; break; }
            const unsigned char d = p[i + k];
            const unsigned char lo = (k == 1) ? lo2 : 0x80, hi = (k == 1) ? hi2 : 0xBF;
            if (d < lo || d > hi) { ok = false; break; }
            cp = (cp << 6) | (d & 0x3F);
        }
        if (!ok || cp < min) { out.push_back(0xFFFD); i += k; continue; }
        out.push_back(cp);
        i += len;
    }
    return out;
}

void utf8_append(std::string& out, uint32_t cp) {
    if (cp < 0x80) out.push_back((char)cp);
    else if (cp < 0x800) { out.push_back((char)(0xC0 | (cp >> 6))); out.push_back((char)(0x80 | (cp & 0x3F))); }
    else if (cp < 0x10000) {
        out.push_back((char)(0xE0 | (cp >> 12))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F)));
    } else {
        out.push_back((char)(0xF0 | (cp >> 18))); out.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
        out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F)));
    }
}

// ---- construction (tokenizer.rs:6-28, 40-72, 85-120) ------------------------------------------------
static const char kSot[] = "<|startoftext|>";
static const char kEot[] = "<|endoftext|>";

Tokenizer::Tokenizer(const std::string& merges_path) {
    // bytes_to_unicode: '!'..'~', 0xA1..0xAC, 0xAE..0xFF keep their code point; the others get 256, 257, ...
    std::vector<int> order;
    std::vector<uint32_t> sym(256, 0);
    auto keep = [](int b) { return (b >= '!' && b <= '~') || (b >= 0xA1 && b <= 0xAC) || (b >= 0xAE && b <= 0xFF); };
    for (int b = 0; b < 256; ++b)
        if (keep(b)) { order.push_back(b); sym[b] = (uint32_t)b; }
    uint32_t next = 256;
    for (int b = 0; b < 256; ++b)
        if (!keep(b)) { order.push_back(b); sym[b] = next++; }
    for (int b = 0; b < 256; ++b) {
        utf8_append(byte_to_sym_[b], sym[b]);
        sym_to_byte_[sym[b]] = (uint8_t)b;
    }

    // load_merges: every line with at least two whitespace-separated words; new() keeps merges[1 .. 48895)
    std::ifstream f(merges_path);
    if (!f) throw Error(SDMI_ERR_IO, "tokenizer: cannot open merges file " + merges_path);
    std::vector<std::pair<std::string, std::string>> merges;
    std::string line;
    while (std::getline(f, line)) {
        std::istringstream ls(line);
        std::string a, b;
        if (ls >> a >> b) merges.emplace_back(a, b);
    }
    const size_t first = 1, last = std::min(merges.size(), (size_t)(49152 - 256 - 2 + 1));
    if (merges.size() < 2) throw Error(SDMI_ERR_IO, "tokenizer: no merges in " + merges_path);

    // construct_vocab: the 256 symbols, the 256 symbols + "</w>", the merges, the two specials
    auto add = [&](const std::string& v) {
        encoder_[v] = (int32_t)decoder_.size();   // HashMap collect: a later duplicate key wins
        decoder_.push_back(v);
    };
    for (int b : order) add(byte_to_sym_[b]);
    for (int b : order) add(byte_to_sym_[b] + "</w>");
    for (size_t i = first; i < last; ++i) {
        add(merges[i].first + merges[i].second);
        ranks_[merges[i].first + " " + merges[i].second] = (int32_t)(i - first);
    }
    add(kSot);
    add(kEot);
    sot_ = encoder_[kSot];
    eot_ = encoder_[kEot];
}

// ---- bpe (tokenizer.rs:122-166) ---------------------------------------------------------------------------
std::vector<std::string> Tokenizer::bpe(const std::vector<std::string>& symbols) const {
    std::vector<std::string> word = symbols;
    word.back() += "</w>";
    if (word.size() < 2) return word;
    for (;;) {
        int best = INT_MAX;
        size_t best_i = 0;
        for (size_t i = 0; i + 1 < word.size(); ++i) {
            auto it = ranks_.find(word[i] + " " + word[i + 1]);
            if (it != ranks_.end() && it->second < best) { best = it->second; best_i = i; }
        }
        if (best == INT_MAX) break;
        const std::string first = word[best_i], second = word[best_i + 1];
        std::vector<std::string> nw;
        size_t i = 0;
        while (i < word.size()) {
            size_t j = i;
            while (j < word.size() && word[j] != first) ++j;
            nw.insert(nw.end(), word.begin() + i, word.begin() + j);
            if (j == word.size()) break;
            i = j;
            if (i + 1 < word.size() && word[i + 1] == second) { nw.push_back(first + second); i += 2; }
            else { nw.push_back(word[i]); i += 1; }
        }
        word.swap(nw);
        if (word.size() == 1) break;
    }
    return word;
}

// ---- encode (tokenizer.rs:168-189) --------------------------------------------------------------------------
// pattern (tokenizer.rs:105), alternatives tried in order at every position, case-insensitively on text
// that is already lower case:
//   <|startoftext|> | <|endoftext|> | 's | 't | 're | 've | 'm | 'll | 'd | \p{L}+ | \p{N} | [^\s\p{L}\p{N}]+
std::vector<int32_t> Tokenizer::encode(const std::string& text) const {
    // text.trim(), whitespace_clean (split_whitespace + join " "), to_lowercase
    std::vector<uint32_t> raw = utf8_decode_lossy(text), cleaned;
    bool pending_space = false;
    for (uint32_t cp : raw) {
        if (unicode_is_space(cp)) { pending_space = !cleaned.empty(); continue; }
        if (pending_space) cleaned.push_back(' ');
        pending_space = false;
        cleaned.push_back(cp);
    }
    const std::vector<uint32_t> s = unicode_lowercase(cleaned);

    // (?i): simple case folding; on lower-cased text the only non-identity fold onto these literals is U+017F -> 's'
    auto fold = [](uint32_t cp) { return cp == 0x17F ? (uint32_t)'s' : cp; };
    auto starts_with = [&](size_t pos, const char* lit) {
        size_t k = 0;
        for (; lit[k]; ++k)
            if (pos + k >= s.size() || fold(s[pos + k]) != (uint32_t)(unsigned char)lit[k]) return (size_t)0;
        return k;
    };
    static const char* const kLiterals[] = {kSot, kEot, "'s", "'t", "'re", "'ve", "'m", "'ll", "'d"};

    std::vector<int32_t> out;
    size_t pos = 0;
    while (pos < s.size()) {
        size_t len = 0;
        for (const char* lit : kLiterals)
            if ((len = starts_with(pos, lit)) != 0) break;
        if (!len) {
            if (unicode_is_letter(s[pos])) {
                while (pos + len < s.size() && unicode_is_letter(s[pos + len])) ++len;
            } else if (unicode_is_number(s[pos])) {
                len = 1;
            } else if (!unicode_is_space(s[pos])) {
                while (pos + len < s.size() && !unicode_is_space(s[pos + len]) && !unicode_is_letter(s[pos + len]) && !unicode_is_number(s[pos + len])) ++len;
            } else {
                ++pos;  // whitespace: no alternative matches here
                continue;
            }
        }
        std::string tok;
        for (size_t k = 0; k < len; ++k) utf8_append(tok, s[pos + k]);
        pos += len;

        if (tok == kSot || tok == kEot) {   // the cache entries of tokenizer.rs:99-102
            out.push_back(encoder_.at(tok));
            continue;
        }
        std::vector<std::string> symbols;   // bytes -> byte_encoder symbols
        symbols.reserve(tok.size());
        for (unsigned char b : tok) symbols.push_back(byte_to_sym_[b]);
        for (const std::string& piece : bpe(symbols)) {
            auto it = encoder_.find(piece);
            if (it == encoder_.end()) throw Error(SDMI_ERR_INVALID, "tokenizer: piece not in the vocabulary");  // reference: panic
            out.push_back(it->second);
        }
    }
    return out;
}

// ---- decode (tokenizer.rs:191-196) --------------------------------------------------------------------------
std::string Tokenizer::decode(const int32_t* ids, size_t n) const {
    std::string text;
    for (size_t i = 0; i < n; ++i) {
        if (ids[i] < 0 || (size_t)ids[i] >= decoder_.size()) throw Error(SDMI_ERR_INVALID, "tokenizer: token id out of range");
        text += decoder_[(size_t)ids[i]];
    }
    std::string bytes;
    for (uint32_t cp : utf8_decode_lossy(text)) {
        auto it = sym_to_byte_.find(cp);
        if (it == sym_to_byte_.end()) throw Error(SDMI_ERR_INVALID, "tokenizer: symbol outside the byte alphabet");  // reference: panic
        bytes.push_back((char)it->second);
    }
    // from_utf8_lossy(...).replace("</w>", " ")
    std::string lossy;
    for (uint32_t cp : utf8_decode_lossy(bytes)) utf8_append(lossy, cp);
    std::string out;
    for (size_t i = 0; i < lossy.size();) {
        if (lossy.compare(i, 4, "</w>") == 0) { out.push_back(' '); i += 4; }
        else out.push_back(lossy[i++]);
    }
    return out;
}

}  // namespace sdmi
