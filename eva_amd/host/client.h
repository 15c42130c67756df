// client.h — the client side: HipPublic::encrypt (SEALPublic::encrypt, seal.cpp:24-102; encoder + encryptor on the
// device when one is present, ciphertexts left resident), HipSecret (SEALSecret::decrypt, seal.cpp:124-146) and
// generate_keys (seal.cpp:174-203).  Included by public_ctx.h.
#pragma once

namespace evahost {

// SEALPublic::encrypt (seal.cpp:24-102)
inline HipValuation HipPublic::encrypt(const Valuation &inputs, const CKKSSignature &sig) {
  const size_t slots = host->N / 2;
  if (sig.vec_size <= 0) throw std::runtime_error("Signature vector size must be positive");
  if (slots < (size_t)sig.vec_size) throw std::runtime_error("Vector size cannot be larger than slot count");
  if (slots % sig.vec_size) throw std::runtime_error("Vector size must exactly divide the slot count");
  HipValuation out;
  SecureRng rng; // a fresh ChaCha20 stream keyed with 256 bits from the OS for this call (csprng.h)
  for (auto &kv : inputs) {
    const auto &v = kv.second;
    if (v.size() != (size_t)sig.vec_size) throw std::runtime_error("Input size does not match program vector size");
    auto it = sig.inputs.find(kv.first);
    if (it == sig.inputs.end()) throw std::out_of_range("No input named " + kv.first + " in the signature");
    const CKKSEncodingInfo &info = it->second;
    if (info.input_type == Type::Cipher || info.input_type == Type::Plain) {
      if ((uint32_t)info.level >= host->k - 1) throw std::runtime_error("Input level exceeds the modulus chain");
      HostPlain pt;
      pt.limbs = host->k - 1 - (uint32_t)info.level;
      pt.scale = std::pow(2.0, (double)info.scale);
      if (info.input_type == Type::Cipher && client_on_device() && device_encodable(v, pt.scale, pt.limbs)) {
        // encoder and encryptor both on the GPU (evah_pt_encode -> evah_encrypt): the plaintext never
        // exists on the host.  Same plaintext as the host encoder bit for bit (tests/test_encode_parity.py)
        // and the same sampler calls in the same order, hence the same ciphertext as every other path
        out.values[kv.first] = encrypt_on_device(nullptr, &v, pt.scale, pt.limbs, rng);
        continue;
      }
      pt.data.resize((size_t)pt.limbs * host->N);
      std::vector<double> vec(slots);
      for (size_t r = 0; r < slots / v.size(); r++) std::copy(v.begin(), v.end(), vec.begin() + r * v.size());
      host->encode_coeff(vec.data(), pt.scale, pt.limbs, pt.data.data());
      if (info.input_type == Type::Cipher && client_on_device()) {
        // device path: the per-limb transforms, the public-key products and the mod-down run on the
        // GPU (evah_encrypt); the host keeps the FP64 encoder and the sampling (same sampler calls,
        // in the same order, as evahost::encrypt — so both paths give the same ciphertext for the
        // same random stream)
        out.values[kv.first] = encrypt_on_device(&pt, nullptr, pt.scale, pt.limbs, rng);
        continue;
      }
      for (uint32_t i = 0; i < pt.limbs; i++) host->ntt(i, pt.data.data() + (size_t)i * host->N);
      if (info.input_type == Type::Cipher) out.values[kv.first] = evahost::encrypt(*host, pk, pt, rng);
      else out.values[kv.first] = std::move(pt);
    } else {
      out.values[kv.first] = v;
    }
  }
  return out;
}

// EVA_DEVICE_CLIENT=0 keeps encrypt on the host; without a HIP device the host path is the only one
// (encrypt, unlike execute(), is client-side work the reference also does on the CPU)
inline bool HipPublic::client_on_device() {
  if (client_device < 0) {
    const char *e = std::getenv("EVA_DEVICE_CLIENT");
    int n = 0;
    client_device = (!e || std::atoi(e) != 0) && evah_device_count(&n) == 0 && n > 0 ? 1 : 0;
  }
  return client_device == 1;
}

// same bound as HipExecutor::device_encodable: every rounded coefficient below 2^62 and inside the modulus
inline bool HipPublic::device_encodable(const std::vector<double> &in, double scale, uint32_t limbs) const {
  const size_t slots = host->N / 2;
  if (std::getenv("EVA_DEVICE_ENCODE") && !std::atoi(std::getenv("EVA_DEVICE_ENCODE"))) return false;
  if (in.empty() || in.size() > slots || slots % in.size()) return false;
  double sum = 0;
  for (double x : in) {
    if (!std::isfinite(x)) return false;
    sum += std::fabs(x);
  }
  const double bound = 2.0 * sum * (double)(slots / in.size()) * scale / (double)host->N;
  const int bits = (int)std::ceil(std::log2(std::max(bound, 1.0))) + 1;
  return bits < 62 && bits < host->total_bits[limbs];
}

// coeff_pt: the host encoder's coefficient-form plaintext, or (null) values: the slot values for the device encoder
inline HostCipher HipPublic::encrypt_on_device(const HostPlain *coeff_pt, const std::vector<double> *values, double scale, uint32_t limbs, SecureRng &rng) {
  ensure_device(false);
  if (!pk_uploaded) {
    chk(evah_client_key_upload(dev->h, EVAH_KEY_PUBLIC, (const uint64_t *)pk.data.data()));
    pk_uploaded = true;
  }
  const uint32_t N = host->N;
  std::vector<int8_t> u, e0, e1, small((size_t)3 * N);
  host->sample_ternary(rng, u);
  host->sample_error(rng, e0);
  host->sample_error(rng, e1);
  std::copy(u.begin(), u.end(), small.begin());
  std::copy(e0.begin(), e0.end(), small.begin() + N);
  std::copy(e1.begin(), e1.end(), small.begin() + 2 * (size_t)N);
  evah_pt *p = nullptr;
  if (coeff_pt) chk(evah_pt_upload_coeff(dev->h, limbs, scale, (const uint64_t *)coeff_pt->data.data(), &p));
  else chk(evah_pt_encode(dev->h, values->data(), (uint32_t)values->size(), limbs, scale, &p));
  evah_ct *c = nullptr;
  int rc = evah_encrypt(dev->h, p, small.data(), &c);
  evah_pt_free(dev->h, p);
  chk(rc);
  HostCipher out;
  out.size = 2;
  out.limbs = limbs;
  out.scale = scale;
  auto handle = std::make_shared<CtHandle>(dev->h, c);
  if (resident) { // stays in HBM; host words on demand
    out.dev = std::make_shared<DeviceResident>(DeviceResident{dev, nullptr, handle, host->N});
    return out;
  }
  out.data.resize((size_t)2 * out.limbs * N);
  out.words_checked = true;
  chk(evah_ct_download(dev->h, c, (uint64_t *)out.data.data()));
  return out;
}

class HipSecret {
public:
  std::shared_ptr<HostContext> host;
  SecretKey sk;
  int device = 0;
  // decrypt + decode on the GPU when one is present (EVA_DEVICE_CLIENT=0: host); the secret key is
  // uploaded once, in NTT form, to a context of its own
  bool on_device() {
    if (state < 0) {
      const char *e = std::getenv("EVA_DEVICE_CLIENT");
      int n = 0;
      state = (!e || std::atoi(e) != 0) && evah_device_count(&n) == 0 && n > 0 ? 1 : 0;
      if (state == 1) {
        // the device state of the key pair (generate_keys shares one holder between both halves), so
        // that the public context's resident results are read in place
        if (!holder->dev) holder->dev = std::make_shared<DeviceCtx>(host->N, host->primes, device);
        dev = holder->dev;
        chk(evah_client_key_upload(dev->h, EVAH_KEY_SECRET, (const uint64_t *)sk.s_ntt.data()));
      }
    }
    return state == 1;
  }
  int state = -1;
  std::shared_ptr<DeviceHolder> holder = std::make_shared<DeviceHolder>();
  std::shared_ptr<DeviceCtx> dev;
  // SEALSecret::decrypt (seal.cpp:124-146)
  Valuation decrypt(const HipValuation &enc, const CKKSSignature &sig) {
    Valuation out;
    for (auto &kv : enc.values) {
      std::vector<double> v;
      if (auto *c = std::get_if<HostCipher>(&kv.second)) {
        if (on_device()) { // dot product with s, inverse transforms, recomposition and the special FFT on the GPU
          if (c->size < 1 || c->size > 3 || c->limbs < 1 || c->limbs > host->k - 1 ||
              (!resident_only(*c) && c->data.size() != (size_t)c->size * c->limbs * host->N) || (c->dev && c->dev->N != host->N))
            throw std::runtime_error("output " + kv.first + ": ciphertext shape does not match its data or the encryption parameters");
          v.resize((size_t)sig.vec_size);
          if (c->dev && c->dev->root == dev) { // resident on this key pair's device state: read in place
            chk(evah_decrypt_decode(dev->h, c->dev->h->h, (uint32_t)sig.vec_size, v.data()));
          } else {
            evah_ct *h = nullptr;
            chk(evah_ct_upload(dev->h, c->size, c->limbs, c->scale, (const uint64_t *)words(*c).data(), &h));
            int rc = evah_decrypt_decode(dev->h, h, (uint32_t)sig.vec_size, v.data());
            evah_ct_free(dev->h, h);
            chk(rc);
          }
          out[kv.first] = std::move(v);
          continue;
        }
        (void)words(*c);
        auto m = decrypt_to_coeff(*host, sk, *c);
        host->decode_coeff(m.data(), c->limbs, c->scale, v);
      } else if (auto *p = std::get_if<HostPlain>(&kv.second)) {
        std::vector<u64> m = p->data;
        for (uint32_t i = 0; i < p->limbs; i++) host->intt(i, m.data() + (size_t)i * host->N);
        host->decode_coeff(m.data(), p->limbs, p->scale, v);
      } else {
        ConstantValue{std::get<std::vector<double>>(kv.second)}.expand_to(v, (size_t)sig.vec_size);
      }
      v.resize((size_t)sig.vec_size);
      out[kv.first] = std::move(v);
    }
    return out;
  }
};

// generateKeys (seal.cpp:174-203): prime chain from bit sizes, secret/public key, one Galois key
// per exact rotation step, relinearization key.
inline std::pair<std::shared_ptr<HipPublic>, std::shared_ptr<HipSecret>>
generate_keys(const CKKSParameters &params, uint64_t seed = 0) {
  std::vector<int> bits(params.prime_bits.begin(), params.prime_bits.end());
  if (bits.size() < 2) throw std::invalid_argument("need at least two primes (data + special)");
  auto primes = evah::coeff_modulus_create(params.poly_modulus_degree, bits);
  auto host = std::make_shared<HostContext>(params.poly_modulus_degree, primes);
  KeyGenerator kg(*host, seed); // seed == 0: keyed from the OS; otherwise the reproducible test hook
  auto pub = std::make_shared<HipPublic>();
  auto sec = std::make_shared<HipSecret>();
  sec->holder = pub->holder; // one device state for the pair: results stay resident from encrypt to decrypt
  pub->host = host;
  pub->pk = kg.public_key();
  pub->relin = kg.relin_key();
  const uint32_t N = host->N, m = 2 * N;
  for (int step : params.rotations) {
    uint32_t elt;
    if (step == 0) elt = m - 1;
    else {
      uint32_t pos = step < 0 ? (uint32_t)(-(int64_t)step) : (uint32_t)step;
      if (pos >= (N >> 1)) throw std::invalid_argument("step count too large");
      uint32_t s = step < 0 ? (N >> 1) - pos : pos;
      elt = 1;
      for (uint32_t i = 0; i < s; i++) elt = (elt * 3u) & (m - 1);
    }
    if (!pub->galois.count(elt)) pub->galois.emplace(elt, kg.galois_key(elt));
  }
  sec->host = host;
  sec->sk = kg.sk;
  return {pub, sec};
}

} // namespace evahost
