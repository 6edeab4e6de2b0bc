// mpk_reader.hpp -- native reader of Burn's NamedMpkFileRecorder<FullPrecisionSettings> records (the `SDv1-4.mpk` the
// reference's README tells every user to download, /root/reference/README.md:13,27; loaded at src/bin/sample/main.rs:27-34).
//
// Layout ASSUMED (burn 0.14.0, the version Cargo.toml:17 pins; UNPINNED against a real file: no Burn build and no record
// exist in this environment -- tests pin the parser against records written by msgpack-python with this layout):
//   file   = rmp_serde "named" encoding (structs are MessagePack MAPS keyed by field name) of
//            BurnRecord { metadata: BurnMetadata{float,int,format,version,settings}, item: <module record> }
//   module = map field -> module | array (Vec<M>, [M; N]) | nil (Option::None, constants: usize / f64 / Ignored<..> /
//            unit modules such as SILU) | ParamSerde
//   ParamSerde = map { "id": str, "param": TensorData }
//   TensorData = map { "bytes": bin (little-endian elements; burn >= 0.14, serde_bytes), "shape": [u64..], "dtype": "F32" }
//              | map { "value": [f32..], "shape": [..] }          (DataSerialize, burn <= 0.13: accepted, copied out)
// Only bytes 0..end of those maps are interpreted; unknown keys are skipped, so extra metadata is harmless.
// Field names are the Rust struct fields (= the dump tree's directory names, src/model/*/load.rs) except:
//   StableDiffusion.diffusion -> "unet", .alpha_cumulative_products -> "alphas_cumprod";
//   GroupNorm / LayerNorm "gamma" / "beta" -> "weight" / "bias"; a bare Param field (CLIP.position_embedding) -> "<field>/weight".
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace sdmi {

struct MpkTensor {
    std::string name;             // dump-tree name, e.g. "unet/input_blocks/rt1/res/conv_in/weight"
    std::vector<int64_t> shape;
    const unsigned char* data;    // fp32 little-endian values: inside the mapping (TensorData) or in `owned` (legacy flavour)
    size_t count;                 // number of elements
    size_t file_offset;           // of `data` in the file (0 for the legacy flavour)
};

class MpkFile {
public:
    explicit MpkFile(const std::string& path);   // maps the file and indexes every tensor; throws sdmi::Error
    ~MpkFile();
    MpkFile(const MpkFile&) = delete;
    MpkFile& operator=(const MpkFile&) = delete;
    const std::vector<MpkTensor>& tensors() const { return tensors_; }
    const std::string& format() const { return format_; }       // metadata.format ("" if absent)
    const std::string& float_type() const { return float_; }    // metadata.float

private:
    void* map_ = nullptr;
    size_t size_ = 0;
    int fd_ = -1;
    std::vector<MpkTensor> tensors_;
    std::vector<std::vector<float>> owned_;
    std::string format_, float_;
    friend struct MpkParser;
};

}  // namespace sdmi
