// sdmi_sample -- the C++ twin of the reference's `sample` binary (src/bin/sample/main.rs:36-125), same argv:
//
//   sdmi_sample <model_type(burn or dump)> <model_name> <unconditional_guidance_scale> <n_diffusion_steps>
//               <prompt> <output_image_name> [device]
//
// It is a pure consumer of the C ABI (include/sdmi.h) -- the same calls the Rust shim (ffi/sdmi.rs) makes:
// tokenizer -> CLIP context -> sample_image -> PNG.  Differences from the reference, all forced:
//   * model_type "burn" reads the NamedMpkFileRecorder<FullPrecisionSettings> record "<model_name>.mpk" natively (the
//     recorder sets the extension itself, main.rs:27-34; layout assumptions: csrc/mpk_reader.hpp); "dump" is the npy tree
//     of python/dump.py (main.rs:94);
//   * device is "hip", "hip:N" or "cuda[N]" (alias, index N); "cpu" / "mps" are refused -- there is no CPU path;
//   * the reference's noise is unseeded; here SDMI_SEED (default 0) seeds the device generator;
//   * the merges file is $SDMI_BPE_VOCAB, default "bpe_simple_vocab_16e6.txt" in the working directory (tokenizer.rs:91);
//   * SDMI_CONFIG="key=value,..." overrides model dimensions (tests use a small model).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/sdmi.h"

static void die(const char* what) {
    std::fprintf(stderr, "%s: %s\n", what, sdmi_last_error());
    std::exit(1);
}

static void apply_overrides(sdmi_config& cfg, const char* spec) {
    std::string s(spec);
    size_t pos = 0;
    while (pos < s.size()) {
        size_t end = s.find(',', pos);
        if (end == std::string::npos) end = s.size();
        const std::string kv = s.substr(pos, end - pos);
        pos = end + 1;
        const size_t eq = kv.find('=');
        if (eq == std::string::npos) continue;
        const std::string k = kv.substr(0, eq);
        const int v = std::atoi(kv.c_str() + eq + 1);
        if (k == "model_channels") cfg.model_channels = v;
        else if (k == "n_head") cfg.n_head = v;
        else if (k == "ctx_dim") cfg.ctx_dim = v;
        else if (k == "latent_h") cfg.latent_h = v;
        else if (k == "latent_w") cfg.latent_w = v;
        else if (k == "vae_ch") cfg.vae_ch = v;
        else if (k == "precision") cfg.precision = v;
        else if (k == "clip_layers") cfg.clip_layers = v;
        else if (k == "clip_heads") cfg.clip_heads = v;
        else if (k == "clip_vocab") cfg.clip_vocab = v;
        else if (k == "clip_ctx") cfg.clip_ctx = v;
        else { std::fprintf(stderr, "SDMI_CONFIG: unknown key %s\n", k.c_str()); std::exit(1); }
    }
}

int main(int argc, char** argv) {
    if (argc != 7 && argc != 8) {
        std::fprintf(stderr, "Usage: %s <model_type(burn or dump)> <model_name> <unconditional_guidance_scale> <n_diffusion_steps> <prompt> <output_image_name> [device(hip, hip:N)]\n", argv[0]);
        return 1;
    }
    const std::string model_type = argv[1], model_name = argv[2], prompt = argv[5], output = argv[6];
    char* endp = nullptr;
    const double scale = std::strtod(argv[3], &endp);
    if (endp == argv[3] || *endp) { std::fprintf(stderr, "Error: Invalid unconditional guidance scale.\n"); return 1; }
    const long long steps = std::strtoll(argv[4], &endp, 10);
    if (endp == argv[4] || *endp || steps < 0) { std::fprintf(stderr, "Error: Invalid number of diffusion steps.\n"); return 1; }

    sdmi_config cfg;
    sdmi_default_config(&cfg);
    if (argc == 8) {
        std::string d = argv[7];
        for (auto& c : d) c = (char)std::tolower((unsigned char)c);
        if (d.rfind("hip", 0) == 0 || d.rfind("cuda", 0) == 0) {
            const size_t digits = d.find_first_of("0123456789");
            cfg.device = digits == std::string::npos ? 0 : std::atoi(d.c_str() + digits);
        } else {
            std::fprintf(stderr, "Unknown device: %s (this build runs on MI355X only: hip or hip:N)\n", argv[7]);
            return 1;
        }
    }
    if (const char* o = std::getenv("SDMI_CONFIG")) apply_overrides(cfg, o);
    if (cfg.clip_layers <= 0) { std::fprintf(stderr, "Error: the sample binary needs the CLIP text encoder (clip_layers > 0)\n"); return 1; }

    std::printf("Loading tokenizer...\n");
    const char* vocab = std::getenv("SDMI_BPE_VOCAB");
    sdmi_tokenizer* tok = nullptr;
    if (sdmi_tokenizer_create(&tok, vocab ? vocab : "bpe_simple_vocab_16e6.txt") != SDMI_OK) die("Error loading tokenizer");

    std::printf("Loading model...\n");
    sdmi_ctx* ctx = nullptr;
    if (sdmi_create(&ctx, &cfg) != SDMI_OK) die("Error creating device context");
    if (model_type == "burn") {
        std::string file = model_name;
        if (file.size() < 4 || file.compare(file.size() - 4, 4, ".mpk") != 0) file += ".mpk";   // FileRecorder::load sets the extension
        if (sdmi_load_weights_mpk(ctx, file.c_str()) != SDMI_OK || sdmi_finalize_weights(ctx) != SDMI_OK) die("Error loading model");
    } else if (sdmi_load_weights_dir(ctx, model_name.c_str()) != SDMI_OK || sdmi_finalize_weights(ctx) != SDMI_OK) {
        die("Error loading model dump");
    }

    // sd.unconditional_context(&tokenizer); sd.context(&tokenizer, prompt)   (main.rs:100-101)
    const int cd = cfg.ctx_dim, cap = cfg.clip_ctx;
    std::vector<float> uncond((size_t)cap * cd), context((size_t)cap * cd);
    int32_t Tu = 0, T = 0;
    if (sdmi_context(ctx, tok, "", uncond.data(), cap, &Tu) != SDMI_OK) die("Error encoding the empty prompt");
    if (sdmi_context(ctx, tok, prompt.c_str(), context.data(), cap, &T) != SDMI_OK) die("Error encoding the prompt");

    std::printf("Sampling image...\n");
    const int H = 8 * cfg.latent_h, W = 8 * cfg.latent_w;
    std::vector<uint8_t> rgb((size_t)H * W * 3);
    const char* seed_env = std::getenv("SDMI_SEED");
    const uint64_t seed = seed_env ? std::strtoull(seed_env, nullptr, 10) : 0;
    if (sdmi_sample_image(ctx, context.data(), 1, T, uncond.data(), Tu, scale, (size_t)steps, nullptr, seed, rgb.data()) != SDMI_OK)
        die("Error sampling image");

    // save_images (main.rs:118-125): "{basepath}{index}.png"
    const std::string path = output + "0.png";
    if (sdmi_write_png(path.c_str(), rgb.data(), W, H) != SDMI_OK) die("Error saving image");

    sdmi_destroy(ctx);
    sdmi_tokenizer_destroy(tok);
    return 0;
}
