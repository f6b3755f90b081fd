// ot_tool.cpp -- TEST INFRASTRUCTURE: libtorch's own serializer as the yardstick for die-e_amd/ot.py.
//
// die-e stores checkpoints and training data as libtorch archives through tch 0.13 (Cargo.toml:10, unvendored):
//   VarStore::save(path)   (src/alphazero/alphazero.rs:263-265)  -> tch's C shim at_save_multi:
//        torch::serialize::OutputArchive a;  for every named variable: a.write(name, tensor, /*is_buffer=*/false);  a.save_to(path)
//   Tensor::save(path)     (alphazero.rs:169-171)                -> at_save = torch::save(tensor, path) = a.write("0", tensor)
//   VarStore::load(path)   (src/alphazero/nnet.rs:109-118)       -> at_load_callback: torch::jit::load(path).named_parameters()
//   Tensor::load(path)     (alphazero.rs:186-198)                -> at_load = torch::load(tensor, path) = InputArchive::read("0")
// This program does exactly those four calls against the libtorch that ships inside the PyTorch wheel of this image, so that the
// archives die-e_amd/ot.py writes are read by libtorch's C++ reader and the archives libtorch's C++ writer produces are read by
// ot.py (tests/test_host_cpu.py; small ones are committed under tests/golden/ot/).  What stays recalled from memory and is NOT
// pinned by this: the variable NAMES tch generates (`weight`, `bias`, `weight__N`, creation order of nnet.rs:62-97).
//
// Tensors travel between this program and the tests in a trivial container ("manifest"): per tensor one text line
//   <name> <dtype: f32|i8|i64> <ndim> <dims...> <byte offset into the .bin file>
// next to a raw little-endian .bin file.
//
//   ot_tool write  <archive.ot> <manifest.txt> <data.bin>     OutputArchive::write per entry, save_to
//   ot_tool read   <archive.ot> <manifest.txt> <data.bin>     jit::load -> named_parameters() (+ named_buffers())
//   ot_tool read0  <archive.ot> <manifest.txt> <data.bin>     torch::load(tensor, path)
//   ot_tool save0  <archive.ot> <manifest.txt> <data.bin>     torch::save(first tensor, path)
#include <torch/script.h>
#include <torch/serialize.h>

#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

namespace {

struct Entry { std::string name; torch::Tensor t; };

torch::ScalarType dtype_of(const std::string& s) {
    if (s == "f32") return torch::kFloat32;
    if (s == "i8") return torch::kInt8;
    if (s == "i64") return torch::kInt64;
    throw std::runtime_error("dtype " + s);
}
std::string name_of(torch::ScalarType t) {
    if (t == torch::kFloat32) return "f32";
    if (t == torch::kInt8) return "i8";
    if (t == torch::kInt64) return "i64";
    throw std::runtime_error("unsupported dtype in archive");
}

std::vector<Entry> read_manifest(const std::string& man, const std::string& bin) {
    std::ifstream m(man);
    std::ifstream b(bin, std::ios::binary);
    if (!m || !b) throw std::runtime_error("cannot open " + man + " / " + bin);
    std::vector<Entry> out;
    std::string line;
    while (std::getline(m, line)) {
        if (line.empty()) continue;
        std::istringstream ls(line);
        std::string name, dt; int nd;
        ls >> name >> dt >> nd;
        std::vector<int64_t> dims(nd);
        for (auto& d : dims) ls >> d;
        long long off; ls >> off;
        torch::Tensor t = torch::empty(dims, torch::TensorOptions().dtype(dtype_of(dt)));
        b.seekg(off);
        b.read((char*)t.data_ptr(), (std::streamsize)t.nbytes());
        if (!b) throw std::runtime_error("short read for " + name);
        out.push_back({name, t});
    }
    return out;
}

void write_manifest(const std::vector<Entry>& es, const std::string& man, const std::string& bin) {
    std::ofstream m(man);
    std::ofstream b(bin, std::ios::binary);
    long long off = 0;
    for (const auto& e : es) {
        const torch::Tensor t = e.t.contiguous().cpu();
        m << e.name << " " << name_of(t.scalar_type()) << " " << t.dim();
        for (auto d : t.sizes()) m << " " << d;
        m << " " << off << "\n";
        b.write((const char*)t.data_ptr(), (std::streamsize)t.nbytes());
        off += (long long)t.nbytes();
    }
}

}  // namespace

int main(int argc, char** argv) {
    if (argc != 5) { std::cerr << "usage: ot_tool write|read|read0|save0 <archive> <manifest> <bin>\n"; return 2; }
    const std::string mode = argv[1], arc = argv[2], man = argv[3], bin = argv[4];
    try {
        if (mode == "write") {                         // tch at_save_multi
            torch::serialize::OutputArchive a;
            for (const auto& e : read_manifest(man, bin)) a.write(e.name, e.t, /*is_buffer=*/false);
            a.save_to(arc);
        } else if (mode == "save0") {                  // tch at_save
            const auto es = read_manifest(man, bin);
            torch::save(es.at(0).t, arc);
        } else if (mode == "read") {                   // tch at_load_callback
            auto module = torch::jit::load(arc);
            std::vector<Entry> es;
            for (const auto& p : module.named_parameters()) es.push_back({p.name, p.value});
            for (const auto& p : module.named_buffers()) es.push_back({p.name, p.value});
            write_manifest(es, man, bin);
        } else if (mode == "read0") {                  // tch at_load
            torch::Tensor t;
            torch::load(t, arc);
            write_manifest({{"0", t}}, man, bin);
        } else {
            std::cerr << "unknown mode " << mode << "\n"; return 2;
        }
    } catch (const std::exception& e) {
        std::cerr << "ot_tool: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
