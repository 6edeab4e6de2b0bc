//! Rust shim over libsdmi.so for Gadersd/stable-diffusion-burn (SOURCE ONLY: there is no Rust
//! toolchain in the build environment, so this file is kept small and obviously correct; it is
//! what a maintainer drops into `src/` next to `src/backend.rs` -- see INTEGRATION.md).
//!
//! It gives `src/bin/sample/main.rs` a `StableDiffusionMi355` with the same four method
//! signatures it already uses on `StableDiffusion<B>` (src/model/stablediffusion/mod.rs:51-67,
//! 69-100, 102-160) but over plain `Vec<f32>` / slices instead of Burn tensors, so the hot path
//! (DDIM + CFG UNet loop, VAE decode) runs in the HIP library.  CLIP and the tokenizer can stay
//! Burn/Rust, or -- when the dump tree holds the clip/ subtree -- run in the library too
//! (`TokenizerMi355`, `StableDiffusionMi355::context`, mirroring stablediffusion/mod.rs:194-210).  Shape errors panic, exactly like the reference's `.unwrap()`s
//! (stablediffusion/mod.rs:86, unet/mod.rs:134); loader errors are `Box<dyn Error>` like
//! `load_stable_diffusion` (stablediffusion/load.rs:16-33).
use std::error::Error;
use std::ffi::{c_char, c_double, c_float, c_int, c_void, CStr, CString};

#[repr(C)]
pub struct SdmiConfig {
    pub device: i32,
    pub model_channels: i32,
    pub n_head: i32,
    pub ctx_dim: i32,
    pub latent_h: i32,
    pub latent_w: i32,
    pub vae_ch: i32,
    pub max_batch: i32,
    pub precision: i32,
    pub clip_layers: i32,
    pub clip_heads: i32,
    pub clip_vocab: i32,
    pub clip_ctx: i32,
    pub reserved: [i32; 3],
}

#[link(name = "sdmi")]
extern "C" {
    fn sdmi_default_config(cfg: *mut SdmiConfig) -> c_int;
    fn sdmi_create(out: *mut *mut c_void, cfg: *const SdmiConfig) -> c_int;
    fn sdmi_destroy(ctx: *mut c_void);
    fn sdmi_last_error() -> *const c_char;
    fn sdmi_set_weight(ctx: *mut c_void, name: *const c_char, data: *const c_float, ndim: i32, dims: *const i64) -> c_int;
    fn sdmi_load_weights_dir(ctx: *mut c_void, dump_dir: *const c_char) -> c_int;
    fn sdmi_load_weights_mpk(ctx: *mut c_void, mpk_path: *const c_char) -> c_int;
    fn sdmi_finalize_weights(ctx: *mut c_void) -> c_int;
    fn sdmi_set_option(ctx: *mut c_void, key: *const c_char, value: *const c_char) -> c_int;
    fn sdmi_create_multi(out: *mut *mut c_void, cfg: *const SdmiConfig, devices: *const i32, n_devices: i32) -> c_int;
    fn sdmi_destroy_multi(m: *mut c_void);
    fn sdmi_multi_load_weights(m: *mut c_void, kind: *const c_char, path: *const c_char) -> c_int;
    fn sdmi_sample_image_sharded(m: *mut c_void, context: *const c_float, t_len: i32, uncond: *const c_float, tu: i32, scale: c_double,
                                 n_steps: usize, n_images: i32, init_latents: *const c_float, seed: u64, rgb_out: *mut u8) -> c_int;
    fn sdmi_unet_forward(ctx: *mut c_void, x: *const c_float, t: i32, context: *const c_float, n: i32, t_len: i32, out: *mut c_float) -> c_int;
    fn sdmi_sample_latent(ctx: *mut c_void, context: *const c_float, n: i32, t_len: i32, uncond: *const c_float, tu: i32,
                          scale: c_double, n_steps: usize, init_latent: *const c_float, seed: u64, latent_out: *mut c_float) -> c_int;
    fn sdmi_decode_latent(ctx: *mut c_void, latent: *const c_float, n: i32, img_out: *mut c_float) -> c_int;
    fn sdmi_latent_to_image(ctx: *mut c_void, latent: *const c_float, n: i32, rgb_out: *mut u8) -> c_int;
    fn sdmi_sample_image(ctx: *mut c_void, context: *const c_float, n: i32, t_len: i32, uncond: *const c_float, tu: i32,
                         scale: c_double, n_steps: usize, init_latent: *const c_float, seed: u64, rgb_out: *mut u8) -> c_int;
    fn sdmi_tokenizer_create(out: *mut *mut c_void, merges_path: *const c_char) -> c_int;
    fn sdmi_tokenizer_destroy(tok: *mut c_void);
    fn sdmi_tokenizer_encode(tok: *const c_void, text: *const c_char, ids: *mut i32, capacity: i32, n_ids: *mut i32) -> c_int;
    fn sdmi_context(ctx: *mut c_void, tok: *const c_void, text: *const c_char, out: *mut c_float, capacity_tokens: i32, t: *mut i32) -> c_int;
    fn sdmi_write_png(path: *const c_char, rgb: *const u8, width: i32, height: i32) -> c_int;
    fn sdmi_qkv_attention(ctx: *mut c_void, q: *const c_float, k: *const c_float, v: *const c_float, mask: *const c_float,
                          mask_ld: i32, n: i32, nq: i32, nk: i32, n_state: i32, n_head: i32, out: *mut c_float) -> c_int;
}

fn last_error() -> String {
    unsafe { CStr::from_ptr(sdmi_last_error()).to_string_lossy().into_owned() }
}

fn check(status: c_int) {
    if status != 0 {
        panic!("libsdmi status {}: {}", status, last_error()); // the reference's hot path is infallible by type
    }
}

/// `SimpleTokenizer` (src/tokenizer.rs:74-196) inside the library.
pub struct TokenizerMi355 {
    tok: *mut c_void,
}

impl TokenizerMi355 {
    /// `SimpleTokenizer::new()` reads "bpe_simple_vocab_16e6.txt" from the working directory (tokenizer.rs:91).
    pub fn new(merges_path: &str) -> Result<Self, Box<dyn Error>> {
        let p = CString::new(merges_path)?;
        let mut tok: *mut c_void = std::ptr::null_mut();
        if unsafe { sdmi_tokenizer_create(&mut tok, p.as_ptr()) } != 0 {
            return Err(last_error().into());
        }
        Ok(Self { tok })
    }

    pub fn encode(&self, text: &str) -> Vec<u32> {
        let t = CString::new(text).expect("prompt contains a NUL byte");
        let cap = 4 * text.len() + 8;
        let mut ids = vec![0i32; cap];
        let mut n = 0i32;
        check(unsafe { sdmi_tokenizer_encode(self.tok, t.as_ptr(), ids.as_mut_ptr(), cap as i32, &mut n) });
        ids[..n as usize].iter().map(|&v| v as u32).collect()
    }
}

impl Drop for TokenizerMi355 {
    fn drop(&mut self) {
        unsafe { sdmi_tokenizer_destroy(self.tok) }
    }
}

pub struct StableDiffusionMi355 {
    ctx: *mut c_void,
    ctx_dim: usize,
    clip_ctx: usize,
    latent: usize, // 4 * h * w
}

impl StableDiffusionMi355 {
    /// `load_stable_diffusion(path, device)` (stablediffusion/load.rs:16-33) for the npy dump tree.
    pub fn load(dump_dir: &str, device: i32) -> Result<Self, Box<dyn Error>> {
        unsafe {
            let mut cfg: SdmiConfig = std::mem::zeroed();
            sdmi_default_config(&mut cfg);
            cfg.device = device;
            let mut ctx: *mut c_void = std::ptr::null_mut();
            if sdmi_create(&mut ctx, &cfg) != 0 {
                return Err(last_error().into());
            }
            let dir = CString::new(dump_dir)?;
            if sdmi_load_weights_dir(ctx, dir.as_ptr()) != 0 || sdmi_finalize_weights(ctx) != 0 {
                let e = last_error();
                sdmi_destroy(ctx);
                return Err(e.into());
            }
            Ok(Self { ctx, ctx_dim: cfg.ctx_dim as usize, clip_ctx: cfg.clip_ctx as usize,
                      latent: 4 * (cfg.latent_h * cfg.latent_w) as usize })
        }
    }

    /// An engine option (`sdmi_set_option`; INTEGRATION.md "Options"), e.g. `("gemm_f32s", "0")` + `("attn_split", "0")` for the fp32 matrix
    /// instruction everywhere, or `("fp8_linear", "0")` for round 2's MXFP8 set.  Unknown keys are an error, not ignored.
    pub fn set_option(&mut self, key: &str, value: &str) -> Result<(), Box<dyn Error>> {
        let (k, v) = (CString::new(key)?, CString::new(value)?);
        if unsafe { sdmi_set_option(self.ctx, k.as_ptr(), v.as_ptr()) } != 0 {
            return Err(last_error().into());
        }
        Ok(())
    }

    /// `load_stable_diffusion_model_file(filename, device)` (src/bin/sample/main.rs:27-34): the Burn
    /// `NamedMpkFileRecorder<FullPrecisionSettings>` record, read natively by the library (the recorder appends ".mpk").
    /// `precision`: 0 fp32 (the reference's arithmetic), 1 bf16, 2 bf16 + MXFP8 convolutions and Linear layers (INTEGRATION.md "Precision").
    pub fn load_record(model_name: &str, device: i32, precision: i32) -> Result<Self, Box<dyn Error>> {
        unsafe {
            let mut cfg: SdmiConfig = std::mem::zeroed();
            sdmi_default_config(&mut cfg);
            cfg.device = device;
            cfg.precision = precision;
            let mut ctx: *mut c_void = std::ptr::null_mut();
            if sdmi_create(&mut ctx, &cfg) != 0 {
                return Err(last_error().into());
            }
            let file = if model_name.ends_with(".mpk") { model_name.to_string() } else { format!("{}.mpk", model_name) };
            let path = CString::new(file)?;
            if sdmi_load_weights_mpk(ctx, path.as_ptr()) != 0 || sdmi_finalize_weights(ctx) != 0 {
                let e = last_error();
                sdmi_destroy(ctx);
                return Err(e.into());
            }
            Ok(Self { ctx, ctx_dim: cfg.ctx_dim as usize, clip_ctx: cfg.clip_ctx as usize,
                      latent: 4 * (cfg.latent_h * cfg.latent_w) as usize })
        }
    }

    /// `context(&tokenizer, text)` (stablediffusion/mod.rs:198-210): `[T, ctx_dim]` row-major, T = tokens + 2.
    /// Needs the clip/ subtree in the dump; `unconditional_context` (:194-196) is `context(tok, "")`.
    pub fn context(&self, tokenizer: &TokenizerMi355, text: &str) -> Vec<f32> {
        let t = CString::new(text).expect("prompt contains a NUL byte");
        let mut out = vec![0f32; self.clip_ctx * self.ctx_dim];
        let mut n_tok = 0i32;
        check(unsafe { sdmi_context(self.ctx, tokenizer.tok, t.as_ptr(), out.as_mut_ptr(), self.clip_ctx as i32, &mut n_tok) });
        out.truncate(n_tok as usize * self.ctx_dim);
        out
    }

    /// `save_images` (src/bin/sample/main.rs:118-125) without the `image` crate.
    pub fn save_images(images: &[Vec<u8>], basepath: &str, width: u32, height: u32) -> Result<(), Box<dyn Error>> {
        for (index, img) in images.iter().enumerate() {
            let p = CString::new(format!("{}{}.png", basepath, index))?;
            if unsafe { sdmi_write_png(p.as_ptr(), img.as_ptr(), width as i32, height as i32) } != 0 {
                return Err(last_error().into());
            }
        }
        Ok(())
    }

    /// One tensor of a Burn record / npy dump, by its dump-tree name (unet/load.rs, autoencoder/load.rs).
    pub fn set_weight(&mut self, name: &str, data: &[f32], dims: &[i64]) -> Result<(), Box<dyn Error>> {
        let n = CString::new(name)?;
        let st = unsafe { sdmi_set_weight(self.ctx, n.as_ptr(), data.as_ptr(), dims.len() as i32, dims.as_ptr()) };
        if st != 0 { Err(last_error().into()) } else { Ok(()) }
    }

    /// `sample_image(context [n,T,768], unconditional_context [Tu,768], scale, n_steps) -> Vec<Vec<u8>>`
    /// (stablediffusion/mod.rs:51-67).  `seed` replaces the reference's unseeded `Tensor::random`.
    pub fn sample_image(&self, context: &[f32], n_batch: usize, unconditional_context: &[f32],
                        unconditional_guidance_scale: f64, n_steps: usize, seed: u64) -> Vec<Vec<u8>> {
        let t = context.len() / (n_batch * self.ctx_dim);
        let tu = unconditional_context.len() / self.ctx_dim;
        assert_eq!(context.len(), n_batch * t * self.ctx_dim);
        let per = self.latent / 4 * 64 * 3; // 512*512*3
        let mut flat = vec![0u8; n_batch * per];
        check(unsafe {
            sdmi_sample_image(self.ctx, context.as_ptr(), n_batch as i32, t as i32, unconditional_context.as_ptr(), tu as i32,
                              unconditional_guidance_scale, n_steps, std::ptr::null(), seed, flat.as_mut_ptr())
        });
        flat.chunks(per).map(|c| c.to_vec()).collect()
    }

    /// `sample_latent` (stablediffusion/mod.rs:102-160) -> [n,4,64,64] row-major.
    pub fn sample_latent(&self, context: &[f32], n_batch: usize, unconditional_context: &[f32],
                         unconditional_guidance_scale: f64, n_steps: usize, init_latent: Option<&[f32]>, seed: u64) -> Vec<f32> {
        let t = context.len() / (n_batch * self.ctx_dim);
        let tu = unconditional_context.len() / self.ctx_dim;
        let mut out = vec![0f32; n_batch * self.latent];
        let x0 = init_latent.map_or(std::ptr::null(), |s| { assert_eq!(s.len(), n_batch * self.latent); s.as_ptr() });
        check(unsafe {
            sdmi_sample_latent(self.ctx, context.as_ptr(), n_batch as i32, t as i32, unconditional_context.as_ptr(), tu as i32,
                               unconditional_guidance_scale, n_steps, x0, seed, out.as_mut_ptr())
        });
        out
    }

    /// `latent_to_image` (stablediffusion/mod.rs:69-100).
    pub fn latent_to_image(&self, latent: &[f32]) -> Vec<Vec<u8>> {
        let n = latent.len() / self.latent;
        assert_eq!(latent.len(), n * self.latent);
        let per = self.latent / 4 * 64 * 3;
        let mut flat = vec![0u8; n * per];
        check(unsafe { sdmi_latent_to_image(self.ctx, latent.as_ptr(), n as i32, flat.as_mut_ptr()) });
        flat.chunks(per).map(|c| c.to_vec()).collect()
    }

    /// `UNet::forward(x, timesteps, context)` (unet/mod.rs:109-143), single shared timestep.
    pub fn unet_forward(&self, x: &[f32], timestep: i32, context: &[f32]) -> Vec<f32> {
        let n = x.len() / self.latent;
        let t = context.len() / (n * self.ctx_dim);
        let mut out = vec![0f32; x.len()];
        check(unsafe { sdmi_unet_forward(self.ctx, x.as_ptr(), timestep, context.as_ptr(), n as i32, t as i32, out.as_mut_ptr()) });
        out
    }

    /// `Autoencoder::decode_latent` (autoencoder/mod.rs:68-71) -> [n,3,512,512].
    pub fn decode_latent(&self, latent: &[f32]) -> Vec<f32> {
        let n = latent.len() / self.latent;
        let mut out = vec![0f32; n * self.latent / 4 * 64 * 3];
        check(unsafe { sdmi_decode_latent(self.ctx, latent.as_ptr(), n as i32, out.as_mut_ptr()) });
        out
    }

    /// The operator seam of the commented-out `trait Backend` (backend.rs:4-84):
    /// `qkv_attention(q, k, v, mask, n_head)` with q [n,nq,c], k/v [n,nk,c].
    pub fn qkv_attention(&self, q: &[f32], k: &[f32], v: &[f32], mask: Option<(&[f32], usize)>,
                         n: usize, nq: usize, nk: usize, n_state: usize, n_head: usize) -> Vec<f32> {
        let mut out = vec![0f32; q.len()];
        let (mp, ld) = mask.map_or((std::ptr::null(), 0), |(m, ld)| (m.as_ptr(), ld));
        check(unsafe {
            sdmi_qkv_attention(self.ctx, q.as_ptr(), k.as_ptr(), v.as_ptr(), mp, ld as i32, n as i32, nq as i32, nk as i32,
                               n_state as i32, n_head as i32, out.as_mut_ptr())
        });
        out
    }
}

impl Drop for StableDiffusionMi355 {
    fn drop(&mut self) {
        unsafe { sdmi_destroy(self.ctx) }
    }
}

/// `sd.sample_image(context, unconditional_context, scale, n_steps)` for `n_images` images of ONE prompt, sharded over the
/// GPUs of a node inside the library (sdmi_create_multi): one weights replica, stream and host thread per device, ONE RCCL
/// broadcast of the packed prompt embedding per call, contiguous image ranges, noise keyed by the global image index
/// (seed + i) -- the result does not depend on the device count.  What `main.rs:104-109` calls when more than one image
/// is wanted.
pub struct StableDiffusionMi355Node {
    m: *mut c_void,
    ctx_dim: usize,
    image_bytes: usize,
}

impl StableDiffusionMi355Node {
    /// kind = "dump" (npy tree, load_stable_diffusion) | "burn" (.mpk record, load_stable_diffusion_model_file)
    pub fn load(kind: &str, path: &str, devices: &[i32], precision: i32) -> Result<Self, Box<dyn Error>> {
        unsafe {
            let mut cfg: SdmiConfig = std::mem::zeroed();
            sdmi_default_config(&mut cfg);
            cfg.precision = precision;
            let mut m: *mut c_void = std::ptr::null_mut();
            if sdmi_create_multi(&mut m, &cfg, devices.as_ptr(), devices.len() as i32) != 0 {
                return Err(last_error().into());
            }
            let (k, p) = (CString::new(kind)?, CString::new(path)?);
            if sdmi_multi_load_weights(m, k.as_ptr(), p.as_ptr()) != 0 {
                let e = last_error();
                sdmi_destroy_multi(m);
                return Err(e.into());
            }
            Ok(Self { m, ctx_dim: cfg.ctx_dim as usize, image_bytes: 3 * 64 * (cfg.latent_h * cfg.latent_w) as usize })
        }
    }

    pub fn sample_image(&self, context: &[f32], unconditional_context: &[f32], unconditional_guidance_scale: f64, n_steps: usize,
                        n_images: usize, seed: u64) -> Vec<Vec<u8>> {
        assert!(context.len() % self.ctx_dim == 0 && unconditional_context.len() % self.ctx_dim == 0, "embedding width");
        let mut rgb = vec![0u8; n_images * self.image_bytes];
        let st = unsafe {
            sdmi_sample_image_sharded(self.m, context.as_ptr(), (context.len() / self.ctx_dim) as i32, unconditional_context.as_ptr(),
                                      (unconditional_context.len() / self.ctx_dim) as i32, unconditional_guidance_scale, n_steps,
                                      n_images as i32, std::ptr::null(), seed, rgb.as_mut_ptr())
        };
        if st != 0 {
            panic!("sdmi_sample_image_sharded: {}", last_error());
        }
        rgb.chunks(self.image_bytes).map(|c| c.to_vec()).collect()
    }
}

impl Drop for StableDiffusionMi355Node {
    fn drop(&mut self) {
        unsafe { sdmi_destroy_multi(self.m) }
    }
}
