// module.cpp — pybind11 module eva_amd._eva: the same Python-visible surface as the reference's
// eva._eva (/root/reference/python/eva/wrapper.cpp:26-246) over the arena IR, the CKKS compiler
// and the MI355X executor.  Submodules: _ckks (compiler), _seal (backend; the name is kept so
// `from eva.seal import generate_keys` keeps working — the backend behind it is libeva_hip.so).
#include <pybind11/numpy.h>
#include <chrono>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include "executor.h"
#include "serialization.h"

namespace py = pybind11;
using namespace evahost;

namespace {

struct PyTerm {
  Program *prog;
  TermId id;
  py::object owner; // keeps the owning Program alive while a handle exists
};

std::vector<TermId> ids_of(const std::vector<PyTerm> &v) {
  std::vector<TermId> out;
  for (auto &t : v) out.push_back(t.id);
  return out;
}

template <class Vec> py::array_t<uint64_t> to_numpy(const Vec &v, std::vector<py::ssize_t> shape) {
  py::array_t<uint64_t> a(shape);
  std::memcpy(a.mutable_data(), v.data(), v.size() * sizeof(u64));
  return a;
}

static int g_num_threads = 1;

} // namespace

PYBIND11_MODULE(_eva, m) {
  m.doc() = "MI355X-native EVA: Python wrapper";

  py::enum_<Op>(m, "Op")
      .value("Undef", Op::Undef).value("Input", Op::Input).value("Output", Op::Output).value("Constant", Op::Constant)
      .value("Negate", Op::Negate).value("Add", Op::Add).value("Sub", Op::Sub).value("Mul", Op::Mul)
      .value("RotateLeftConst", Op::RotateLeftConst).value("RotateRightConst", Op::RotateRightConst)
      .value("Relinearize", Op::Relinearize).value("ModSwitch", Op::ModSwitch).value("Rescale", Op::Rescale)
      .value("Encode", Op::Encode);
  py::enum_<Type>(m, "Type")
      .value("Undef", Type::Undef).value("Cipher", Type::Cipher).value("Raw", Type::Raw).value("Plain", Type::Plain);

  py::class_<PyTerm>(m, "Term", "Native Term handle")
      .def_property_readonly("op", [](const PyTerm &t) { return t.prog->at(t.id).op; }, "The operation performed by this term")
      .def_property_readonly("index", [](const PyTerm &t) { return t.id; });

  py::class_<Program>(m, "Program", "Native Program class")
      .def(py::init<std::string, uint64_t>(), py::arg("name"), py::arg("vec_size"))
      .def_property("name", &Program::name, &Program::set_name, "The name of this program")
      .def_property_readonly("vec_size", &Program::vec_size, "The number of elements for all vectors in this program")
      .def_property_readonly("inputs", [](py::object self) {
        Program &p = self.cast<Program &>();
        std::unordered_map<std::string, PyTerm> out;
        for (auto &kv : p.inputs()) out.emplace(kv.first, PyTerm{&p, kv.second, self});
        return out;
      }, "A dictionary from input names to terms")
      .def_property_readonly("outputs", [](py::object self) {
        Program &p = self.cast<Program &>();
        std::unordered_map<std::string, PyTerm> out;
        for (auto &kv : p.outputs()) out.emplace(kv.first, PyTerm{&p, kv.second, self});
        return out;
      }, "A dictionary from output names to terms")
      .def("set_output_ranges", [](Program &p, uint32_t range) {
        for (auto &kv : p.outputs()) { p.at(kv.second).has_range = true; p.at(kv.second).range = range; }
      }, py::arg("range"), "Sets the range (in bits) all outputs must accommodate")
      .def("set_input_scales", [](Program &p, uint32_t scale) {
        for (TermId s : p.sources()) { p.at(s).has_encode_scale = true; p.at(s).encode_scale = scale; }
      }, py::arg("scale"), "Sets the scale (in bits) all inputs and constants are encoded at")
      .def("to_DOT", &Program::to_dot)
      .def("_make_term", [](py::object self, Op op, const std::vector<PyTerm> &operands) { Program &p = self.cast<Program &>(); return PyTerm{&p, p.make_term(op, ids_of(operands)), self}; })
      .def("_make_left_rotation", [](py::object self, const PyTerm &t, int32_t s) { Program &p = self.cast<Program &>(); return PyTerm{&p, p.make_left_rotation(t.id, s), self}; })
      .def("_make_right_rotation", [](py::object self, const PyTerm &t, int32_t s) { Program &p = self.cast<Program &>(); return PyTerm{&p, p.make_right_rotation(t.id, s), self}; })
      .def("_make_dense_constant", [](py::object self, std::vector<double> v) { Program &p = self.cast<Program &>(); return PyTerm{&p, p.make_dense_constant(std::move(v)), self}; })
      .def("_make_uniform_constant", [](py::object self, double v) { Program &p = self.cast<Program &>(); return PyTerm{&p, p.make_uniform_constant(v), self}; })
      .def("_make_input", [](py::object self, const std::string &name, Type t) { Program &p = self.cast<Program &>(); return PyTerm{&p, p.make_input(name, t), self}; })
      .def("_make_output", [](py::object self, const std::string &name, const PyTerm &t) { Program &p = self.cast<Program &>(); return PyTerm{&p, p.make_output(name, t.id), self}; })
      // introspection used by the parity tests: the live DAG in topological order
      .def("_dump", [](Program &p) {
        py::list out;
        for (TermId t : p.topo_order()) {
          const Term &x = p.at(t);
          py::dict d;
          d["id"] = t;
          d["op"] = x.op;
          d["operands"] = x.operands;
          if (x.has_rotation) d["rotation"] = x.rotation;
          if (x.has_rescale_divisor) d["rescale_divisor"] = x.rescale_divisor;
          if (x.has_type) d["type"] = x.type_attr;
          if (x.has_range) d["range"] = x.range;
          if (x.has_encode_scale) d["encode_scale"] = x.encode_scale;
          if (x.has_encode_level) d["encode_level"] = x.encode_level;
          if (x.constant) d["constant"] = x.constant->values;
          out.append(d);
        }
        return out;
      });

  m.def("evaluate", &evaluate, py::arg("program"), py::arg("inputs"), "Evaluate the program without homomorphic encryption (reference semantics)");
  // serialization (wrapper.cpp:110-116)
  // Program / CKKSParameters / CKKSSignature: the reference's protobuf wire format by default (files
  // interchange with microsoft/EVA); format="native" = this repo's container
  auto want_wire = [](const std::string &format) {
    if (format == "eva") return true;
    if (format == "native") return false;
    throw std::invalid_argument("format must be 'eva' or 'native'");
  };
  m.def("save", [want_wire](const Program &o, const std::string &path, const std::string &format) {
    if (want_wire(format)) save_wire_to_file("Program", wire::encode(o), path);
    else save_to_file(Kind::Program, o, path);
  }, py::arg("obj"), py::arg("path"), py::arg("format") = "eva");
  m.def("save", [want_wire](const CKKSParameters &o, const std::string &path, const std::string &format) {
    if (want_wire(format)) save_wire_to_file("CKKSParameters", wire::encode(o), path);
    else save_to_file(Kind::Parameters, o, path);
  }, py::arg("obj"), py::arg("path"), py::arg("format") = "eva");
  m.def("save", [want_wire](const CKKSSignature &o, const std::string &path, const std::string &format) {
    if (want_wire(format)) save_wire_to_file("CKKSSignature", wire::encode(o), path);
    else save_to_file(Kind::Signature, o, path);
  }, py::arg("obj"), py::arg("path"), py::arg("format") = "eva");
  // valuations and key contexts: this repo's container by default; format="seal" (or "seal+zlib" / "seal+zstd") = the
  // reference's protobuf messages around SEAL's binary object format (seal.proto, seal_format.h)
  auto seal_compr = [](const std::string &format, bool &seal) {
    seal = true;
    if (format == "seal") return sealfmt::None;
    if (format == "seal+zlib") return sealfmt::Zlib;
    if (format == "seal+zstd") return sealfmt::Zstd;
    seal = false;
    if (format == "native") return sealfmt::None;
    throw std::invalid_argument("format must be 'native', 'seal', 'seal+zlib' or 'seal+zstd'");
  };
  m.def("save", [seal_compr](const HipValuation &o, const std::string &path, const std::string &format) {
    bool seal = false;
    const sealfmt::Compr c = seal_compr(format, seal);
    if (seal) save_wire_to_file("SEALValuation", sealfmt::encode_valuation(o, c), path);
    else save_to_file(Kind::Valuation, o, path);
  }, py::arg("obj"), py::arg("path"), py::arg("format") = "native");
  m.def("save", [seal_compr](const HipPublic &o, const std::string &path, const std::string &format) {
    bool seal = false;
    const sealfmt::Compr c = seal_compr(format, seal);
    if (seal) save_wire_to_file("SEALPublic", sealfmt::encode_public(o, c), path);
    else save_to_file(Kind::Public, o, path);
  }, py::arg("obj"), py::arg("path"), py::arg("format") = "native");
  m.def("save", [seal_compr](const HipSecret &o, const std::string &path, const std::string &format) {
    bool seal = false;
    const sealfmt::Compr c = seal_compr(format, seal);
    if (seal) save_wire_to_file("SEALSecret", sealfmt::encode_secret(o, c), path);
    else save_to_file(Kind::Secret, o, path);
  }, py::arg("obj"), py::arg("path"), py::arg("format") = "native");
  // test hooks of the SEAL object format
  m.def("_blake2b_256", [](py::bytes data) {
    const std::string s = data;
    uint8_t out[32];
    sealfmt::blake2b(out, 32, (const uint8_t *)s.data(), s.size());
    return py::bytes((const char *)out, 32);
  });
  m.def("_seal_zstd_available", []() { return sealfmt::ZstdLib::get().ok(); });
  // one SEAL object (the bytes of a SEALObject.data) from raw words: what tests/golden/export_seal_vectors.py hands to
  // tools/seal_parity.cpp, which loads them with SEAL itself and compares SEAL's own save() with them
  m.def("_seal_blob", [](const std::string &kind, uint32_t N, const std::vector<uint64_t> &primes,
                         py::array_t<uint64_t, py::array::c_style | py::array::forcecast> data, double scale) {
    HostContext h(N, std::vector<u64>(primes.begin(), primes.end()));
    const u64 *d = (const u64 *)data.data();
    auto shape_is = [&](std::initializer_list<py::ssize_t> want) {
      if ((size_t)data.ndim() != want.size()) throw std::invalid_argument("wrong number of dimensions for a SEAL " + kind);
      size_t i = 0;
      for (py::ssize_t w : want) {
        if (w >= 0 && data.shape(i) != w) throw std::invalid_argument("wrong shape for a SEAL " + kind);
        i++;
      }
    };
    std::string out;
    if (kind == "parms") out = sealfmt::parms_obj(h, sealfmt::None);
    else if (kind == "ciphertext") { shape_is({-1, -1, (py::ssize_t)N}); out = sealfmt::ciphertext_obj(h, (uint32_t)data.shape(0), (uint32_t)data.shape(1), scale, d, sealfmt::None); }
    else if (kind == "plaintext") { shape_is({-1, (py::ssize_t)N}); out = sealfmt::plaintext_obj(h, (uint32_t)data.shape(0), scale, d, sealfmt::None); }
    else if (kind == "public_key") { shape_is({2, (py::ssize_t)h.k, (py::ssize_t)N}); out = sealfmt::public_key_obj(h, d, sealfmt::None); }
    else if (kind == "secret_key") { shape_is({(py::ssize_t)h.k, (py::ssize_t)N}); out = sealfmt::secret_key_obj(h, d, sealfmt::None); }
    else if (kind == "relin_keys") {
      shape_is({-1, 2, (py::ssize_t)h.k, (py::ssize_t)N});
      SwitchKey k;
      k.n_digits = (uint32_t)data.shape(0);
      k.data.assign(d, d + data.size());
      out = sealfmt::kswitch_obj(h, 1, {{0, &k}}, sealfmt::None);
    } else throw std::invalid_argument("unknown SEAL object kind " + kind);
    return py::bytes(out);
  }, py::arg("kind"), py::arg("N"), py::arg("primes"), py::arg("data"), py::arg("scale") = 1.0);
  m.def("_seal_galois_blob", [](uint32_t N, const std::vector<uint64_t> &primes, const std::map<uint32_t, py::array_t<uint64_t, py::array::c_style | py::array::forcecast>> &keys) {
    HostContext h(N, std::vector<u64>(primes.begin(), primes.end()));
    std::map<uint64_t, SwitchKey> own;
    for (auto &kv : keys) {
      if (kv.second.ndim() != 4 || kv.second.shape(1) != 2 || kv.second.shape(2) != (py::ssize_t)h.k || kv.second.shape(3) != (py::ssize_t)N || !(kv.first & 1) || kv.first >= 2 * N)
        throw std::invalid_argument("wrong shape / element for a SEAL Galois key");
      SwitchKey k;
      k.n_digits = (uint32_t)kv.second.shape(0);
      k.data.assign((const u64 *)kv.second.data(), (const u64 *)kv.second.data() + kv.second.size());
      own.emplace((uint64_t)(kv.first - 1) / 2, std::move(k));
    }
    std::map<uint64_t, const SwitchKey *> slots;
    for (auto &kv : own) slots.emplace(kv.first, &kv.second);
    return py::bytes(sealfmt::kswitch_obj(h, N, slots, sealfmt::None));
  }, py::arg("N"), py::arg("primes"), py::arg("keys"));
  m.def("load", [](const std::string &path) -> py::object {
    KnownType k = load_from_file(path);
    if (auto *p = std::get_if<std::unique_ptr<Program>>(&k)) return py::cast(std::move(*p));
    if (auto *p = std::get_if<CKKSParameters>(&k)) return py::cast(*p);
    if (auto *p = std::get_if<CKKSSignature>(&k)) return py::cast(*p);
    if (auto *p = std::get_if<HipValuation>(&k)) return py::cast(std::move(*p));
    if (auto *p = std::get_if<std::shared_ptr<HipPublic>>(&k)) return py::cast(*p);
    return py::cast(std::get<std::shared_ptr<HipSecret>>(k));
  }, py::arg("path"), "Load a previously saved object (same class as was saved)");

  m.def("set_num_threads", [](int n) { if (n < 1) throw std::invalid_argument("num_threads must be positive"); g_num_threads = n; }, py::arg("num_threads"),
        "Node-level parallelism of execute(): contexts made by generate_keys() afterwards spread independent DAG nodes over min(n, 8) "
        "issue queues (HIP streams) — the GPU counterpart of the reference's Galois worker threads; 1 (the default) = one in-order queue");
  struct GaloisGuard {};
  py::class_<GaloisGuard>(m, "_GaloisGuard").def(py::init());

  // ---- CKKS compiler
  py::module mckks = m.def_submodule("_ckks", "CKKS compiler");
  py::class_<CKKSCompiler>(mckks, "CKKSCompiler")
      .def(py::init(), "Create a compiler with the default config")
      .def(py::init([](const std::unordered_map<std::string, std::string> &cfg) { return CKKSCompiler(CKKSConfig(cfg)); }), py::arg("config"))
      .def("compile", [](CKKSCompiler &c, Program &p) {
        auto r = c.compile(p);
        return py::make_tuple(py::cast(std::move(std::get<0>(r))), std::get<1>(r), std::get<2>(r));
      }, py::arg("program"));
  py::class_<CKKSParameters>(mckks, "CKKSParameters", "Abstract encryption parameters for CKKS")
      .def(py::init([](std::vector<uint32_t> bits, std::set<int> rot, uint32_t n) { CKKSParameters p; p.prime_bits = bits; p.rotations = rot; p.poly_modulus_degree = n; return p; }),
           py::arg("prime_bits"), py::arg("rotations"), py::arg("poly_modulus_degree"))
      .def_readwrite("prime_bits", &CKKSParameters::prime_bits)
      .def_readwrite("rotations", &CKKSParameters::rotations)
      .def_readwrite("poly_modulus_degree", &CKKSParameters::poly_modulus_degree);
  py::class_<CKKSEncodingInfo>(mckks, "CKKSEncodingInfo")
      .def_readonly("input_type", &CKKSEncodingInfo::input_type)
      .def_readonly("scale", &CKKSEncodingInfo::scale)
      .def_readonly("level", &CKKSEncodingInfo::level);
  py::class_<CKKSSignature>(mckks, "CKKSSignature")
      .def_readonly("vec_size", &CKKSSignature::vec_size)
      .def_readonly("inputs", &CKKSSignature::inputs);

  // ---- backend
  py::module mseal = m.def_submodule("_seal", "MI355X CKKS execution backend (drop-in for eva._eva._seal)");
  // devices / shard: several GPUs behind one execute() (eva_amd/host/multi_device.h); the defaults come from
  // EVA_NUM_GPUS / EVA_DEVICES / EVA_SHARD.  set_num_threads(n) — the reference's size of the parallel
  // traversal (wrapper.cpp:128-137) — is the number of issue queues independent DAG nodes are spread over.
  mseal.def("generate_keys", [](const CKKSParameters &p, uint64_t seed, py::object devices, py::object shard) {
    auto kp = generate_keys(p, seed);
    if (!devices.is_none()) {
      kp.first->devices = devices.cast<std::vector<int>>();
      // the key pair's own device state (inputs, constants, outputs; the secret half decrypts there) is member 0
      if (!kp.first->devices.empty()) kp.first->device = kp.second->device = evahost::physical_device(kp.first->devices[0]);
    }
    if (!shard.is_none()) kp.first->shard_mode = shard.cast<std::string>();
    if (g_num_threads > 1) kp.first->num_queues = std::min(g_num_threads, 8);
    return kp;
  }, py::arg("abstract_params"), py::arg("seed") = 0, py::arg("devices") = py::none(), py::arg("shard") = py::none());
  py::class_<HipValuation>(mseal, "SEALValuation", "Inputs or outputs of execute(): ciphertexts, plaintexts or raw vectors")
      .def(py::init<>())
      .def("_set_cipher", [](HipValuation &v, const std::string &name, py::array_t<uint64_t, py::array::c_style | py::array::forcecast> data, double scale) {
        if (data.ndim() != 3) throw std::invalid_argument("cipher data must be [size][limbs][N]");
        HostCipher c;
        c.size = (uint32_t)data.shape(0);
        c.limbs = (uint32_t)data.shape(1);
        c.scale = scale;
        c.data.assign((const u64 *)data.data(), (const u64 *)data.data() + data.size());
        v.values[name] = std::move(c);
      })
      .def("_set_plain", [](HipValuation &v, const std::string &name, py::array_t<uint64_t, py::array::c_style | py::array::forcecast> data, double scale) {
        if (data.ndim() != 2) throw std::invalid_argument("plain data must be [limbs][N]");
        HostPlain p;
        p.limbs = (uint32_t)data.shape(0);
        p.scale = scale;
        p.data.assign((const u64 *)data.data(), (const u64 *)data.data() + data.size());
        v.values[name] = std::move(p);
      })
      .def("_set_raw", [](HipValuation &v, const std::string &name, std::vector<double> data) { v.values[name] = std::move(data); })
      // a valuation assembled by hand or loaded from this repo's container carries no encryption parameters; the SEAL
      // wire format needs them (SEALValuation::params, seal.h:23-27)
      .def("_set_params", [](HipValuation &v, const HipPublic &p) { v.params = p.host; }, py::arg("public_ctx"))
      .def("names", [](const HipValuation &v) { std::vector<std::string> n; for (auto &kv : v.values) n.push_back(kv.first); return n; })
      .def("is_resident", [](const HipValuation &v, const std::string &name) {
        auto it = v.values.find(name);
        if (it == v.values.end()) throw std::out_of_range("No value named " + name);
        auto *c = std::get_if<HostCipher>(&it->second);
        return c && c->dev != nullptr;
      }, py::arg("name"), "True while the ciphertext lives in HBM (a handle of the device context that produced it)")
      .def("on_host", [](const HipValuation &v, const std::string &name) {
        auto it = v.values.find(name);
        if (it == v.values.end()) throw std::out_of_range("No value named " + name);
        auto *c = std::get_if<HostCipher>(&it->second);
        return !c || !c->data.empty();
      }, py::arg("name"), "True when host words of the value exist (always, for plaintexts and raw vectors)")
      .def("to_host", [](HipValuation &v, bool drop_device) {
        for (auto &kv : v.values)
          if (auto *c = std::get_if<HostCipher>(&kv.second)) {
            (void)words(*c);
            if (drop_device) c->dev.reset();
          }
      }, py::arg("drop_device") = false, "Download every device-resident ciphertext (waits for it); drop_device releases the HBM copies")
      // (kind, size, limbs, scale, data) — raw residues for the bit-exact parity tests
      .def("get", [](const HipValuation &v, const std::string &name) -> py::object {
        auto it = v.values.find(name);
        if (it == v.values.end()) throw std::out_of_range("No value named " + name);
        if (auto *c = std::get_if<HostCipher>(&it->second)) {
          const CipherWords &w = words(*c); // a device-resident value is downloaded (and kept) here
          py::ssize_t n = (py::ssize_t)(w.size() / ((size_t)c->size * c->limbs));
          return py::make_tuple("cipher", c->size, c->limbs, c->scale, to_numpy(w, {(py::ssize_t)c->size, (py::ssize_t)c->limbs, n}));
        }
        if (auto *p = std::get_if<HostPlain>(&it->second)) {
          py::ssize_t n = (py::ssize_t)(p->data.size() / p->limbs);
          return py::make_tuple("plain", 1, p->limbs, p->scale, to_numpy(p->data, {(py::ssize_t)p->limbs, n}));
        }
        return py::make_tuple("raw", 0, 0, 1.0, py::cast(std::get<std::vector<double>>(it->second)));
      });
  py::class_<HipPublic, std::shared_ptr<HipPublic>>(mseal, "SEALPublic", "Public context: encryption and execution on the MI355X")
      .def("encrypt", [](HipPublic &p, const Valuation &inputs, const CKKSSignature &sig) {
        HipValuation v = p.encrypt(inputs, sig);
        v.params = p.host;
        return v;
      }, py::arg("inputs"), py::arg("signature"))
      .def("execute", [](HipPublic &p, Program &program, const HipValuation &inputs) {
        HipValuation v = p.execute(program, inputs);
        v.params = p.host;
        return v;
      }, py::arg("program"), py::arg("inputs"))
      .def("execute_batch", [](HipPublic &p, Program &program, const std::vector<const HipValuation *> &inputs) {
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<HipValuation> out = p.execute_batch(program, inputs);
        for (HipValuation &v : out) v.params = p.host;
        if (std::getenv("EVA_BATCH_TIMING"))
          std::fprintf(stderr, "EVA: execute_batch binding: %.3f ms inside (before the results become Python objects)\n",
                       std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        return out;
      }, py::arg("program"), py::arg("inputs"),
           "execute() for a list of independent input valuations of one program; instances run batch_chunk at a time as batched device handles")
      .def_readwrite("library_scheduler", &HipPublic::library_scheduler, "run the encrypted part of a program as one evah_execute (default) instead of the node-by-node host walk")
      .def_readwrite("batch_chunk", &HipPublic::batch_chunk, "instances per batched device handle in execute_batch (1..64)")
      .def_readwrite("batch_ramp", &HipPublic::batch_ramp, "execute_batch: quarter / three-quarter sized groups at both ends of the batch, so the pipeline fills and drains on small copies (EVA_BATCH_RAMP)")
      .def_readwrite("batch_balance", &HipPublic::batch_balance, "execute_batch: groups of (nearly) equal size instead of full groups and a remainder (EVA_BATCH_BALANCE)")
      .def_readwrite("batch_depth", &HipPublic::batch_depth, "groups in flight in execute_batch = issue queues it rotates over (2..8; 0 = three on resident valuations, four on host valuations; EVA_BATCH_DEPTH)")
      .def_readwrite("device", &HipPublic::device)
      .def_readwrite("free_eagerly", &HipPublic::free_eagerly)
      .def_readwrite("use_graphs", &HipPublic::use_graphs, "replay repeated executions of one program from a captured hipGraph")
      .def("_graph_plans", [](HipPublic &p) { return p.graph_plan_count(); }, "captured plans alive (a program found busy by the next call has two)")
      .def_readwrite("twin_plans", &HipPublic::twin_plans, "a program whose replay is found busy by the next execute() gets a second captured plan on its own queue; calls then go to whichever is idle (EVA_GRAPH_TWIN)")
      .def("drop_graphs", &HipPublic::drop_graphs)
      .def_readwrite("devices", &HipPublic::devices, "device index per member of the multi-GPU modes (a repeated index = several contexts on one GPU)")
      .def_readwrite("shard_mode", &HipPublic::shard_mode, "'' (one device) | 'subdag' | 'limb' | 'dag' — how execute() / execute_batch() use `devices`")
      .def_readonly("last_subdag_plan", &HipPublic::last_subdag_plan, "(member, ops) per piece of the last sub-DAG split: prefix, components..., suffix")
      .def("key_bytes", &HipPublic::key_bytes, "HBM bytes of evaluation keys: one entry per limb shard (when limb-sharded), then the whole keys on the context's own device (0 if never uploaded)")
      .def("set_limb_dist", [](HipPublic &p, uint32_t rank, uint32_t world, py::object all_gather, py::object broadcast, py::object sum_host,
                               uintptr_t stream) {
        // the exchange steps of limb sharding across processes, as callbacks into torch.distributed (RCCL); called
        // from execute() with the GIL held
        p.limb_rank = rank;
        p.limb_world = world;
        p.limb_stream = stream;
        p.limb_hooks.all_gather = [all_gather](void *ptr, size_t chunk) { all_gather((uintptr_t)ptr, chunk); };
        p.limb_hooks.broadcast = [broadcast](void *ptr, size_t words, uint32_t owner) { broadcast((uintptr_t)ptr, words, owner); };
        p.limb_hooks.sum_host = [sum_host](evahost::u64 *w, size_t n) {
          sum_host(py::array_t<uint64_t>({(py::ssize_t)n}, {(py::ssize_t)sizeof(uint64_t)}, (const uint64_t *)w, py::capsule(w, [](void *) {})));
        };
        p.shard_mode = "limb";
      }, py::arg("rank"), py::arg("world"), py::arg("all_gather"), py::arg("broadcast"), py::arg("sum_host"), py::arg("stream") = 0,
           "limb sharding across processes: this context is shard `rank` of `world`; all_gather(ptr, chunk_words), "
           "broadcast(ptr, words, owner) act in place on device buffers, sum_host(array) all-reduces a host uint64 array")
      .def_readonly("last_exchanged_words", &HipPublic::last_exchanged_words, "uint64 words moved between shards by the last limb-sharded execute()")
      .def_readonly("last_exchange_launches", &HipPublic::last_exchange_launches,
                    "launches those words took: one per receiving shard per exchange step (evah_buf_gather)")
      .def_readwrite("resident", &HipPublic::resident, "keep valuations in HBM: encrypt/execute return device handles and execute does not wait for the GPU (EVA_RESIDENT=0: host valuations)")
      .def_readwrite("graph_copy_limit", &HipPublic::graph_copy_limit, "device-resident inputs above this many bytes are walked eagerly instead of copied into a captured graph's slots")
      .def("synchronize", &HipPublic::synchronize, "wait for everything this context has enqueued")
      .def("transfer_stats", [](HipPublic &p) {
        auto st = p.transfer_stats();
        py::dict d;
        d["ct_uploads"] = st[0]; d["ct_downloads"] = st[1]; d["pt_uploads"] = st[2]; d["pt_downloads"] = st[3];
        d["h2d_bytes"] = st[4]; d["d2h_bytes"] = st[5];
        return d;
      }, "ciphertext / plaintext transfers across the host boundary since the device context was created")
      .def("profile", &HipPublic::profile, py::arg("on"), "HIP-event brackets around every kernel launch of this context's issue queues, by kernel class")
      .def("profile_reset", &HipPublic::profile_reset)
      .def("profile_get", [](HipPublic &p) {
        py::dict d;
        for (auto &kv : p.profile_get()) d[py::str(kv.first)] = py::make_tuple(kv.second.first, kv.second.second);
        return d;
      }, "{kernel class: (launches, total ms)} since the last profile_reset(), summed over the issue queues")
      .def_readonly("last_timing", &HipPublic::last_timing, "ms of the last execute(): (input upload, DAG enqueue on the host, drain + output download)")
      .def_readwrite("num_queues", &HipPublic::num_queues, "HIP streams independent DAG nodes are spread over")
      .def_property_readonly("poly_modulus_degree", [](const HipPublic &p) { return p.host->N; })
      .def_property_readonly("primes", [](const HipPublic &p) { return std::vector<uint64_t>(p.host->primes.begin(), p.host->primes.end()); })
      // host FP64 encoder + host NTT: the plaintext an Encode node produces (tests pin the
      // integer path "from the encoded plaintext onward")
      .def("_encode", [](const HipPublic &p, const std::vector<double> &values, uint32_t scale_bits, uint32_t level) {
        const HostContext &h = *p.host;
        const size_t slots = h.N / 2;
        if (values.empty() || slots % values.size()) throw std::runtime_error("Size must exactly divide slots");
        if (level >= h.k - 1) throw std::runtime_error("Encode level exceeds the modulus chain");
        const uint32_t limbs = h.k - 1 - level;
        std::vector<double> vec;
        for (size_t r = slots / values.size(); r > 0; --r) vec.insert(vec.end(), values.begin(), values.end());
        std::vector<u64> out((size_t)limbs * h.N);
        h.encode_coeff(vec.data(), std::pow(2.0, (double)scale_bits), limbs, out.data());
        for (uint32_t i = 0; i < limbs; i++) h.ntt(i, out.data() + (size_t)i * h.N);
        return to_numpy(out, {(py::ssize_t)limbs, (py::ssize_t)h.N});
      }, py::arg("values"), py::arg("scale_bits"), py::arg("level"))
      .def("relin_key", [](const HipPublic &p) {
        return to_numpy(p.relin.data, {(py::ssize_t)p.relin.n_digits, 2, (py::ssize_t)p.host->k, (py::ssize_t)p.host->N});
      })
      .def("public_key", [](const HipPublic &p) { return to_numpy(p.pk.data, {2, (py::ssize_t)p.host->k, (py::ssize_t)p.host->N}); })
      .def("galois_keys", [](const HipPublic &p) {
        py::dict d;
        for (auto &kv : p.galois)
          d[py::int_(kv.first)] = to_numpy(kv.second.data, {(py::ssize_t)kv.second.n_digits, 2, (py::ssize_t)p.host->k, (py::ssize_t)p.host->N});
        return d;
      });
  py::class_<HipSecret, std::shared_ptr<HipSecret>>(mseal, "SEALSecret", "Secret context: decryption. Holds the secret key.")
      .def("decrypt", &HipSecret::decrypt, py::arg("enc_outputs"), py::arg("signature"))
      .def_readwrite("device", &HipSecret::device, "device of the secret half's state when it is not shared with a public context")
      // test hook (as relin_key() on the public side): the secret key under every key prime, NTT form [k][N]
      .def("_secret_key_ntt", [](const HipSecret &s) { return to_numpy(s.sk.s_ntt, {(py::ssize_t)s.host->k, (py::ssize_t)s.host->N}); });
}
