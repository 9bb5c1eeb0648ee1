// Fixture / mesh I/O of the facade: the CSV dialect of the reference's data files and the loader of its mesh directories.
//   CSVReader<T>::parse_file   fdaPDE/utils/IO/csv_reader.h:75-117  (dense files: one header row, first column = row index, blanks and
//                              double quotes dropped from every token, NA / NaN / nan -> quiet NaN)
//   MeshLoader<M, N>           test/src/utils/mesh_loader.h:62-84     (points.csv, elements.csv, boundary.csv, edges.csv, neigh.csv;
//                              ids are 1-based in the files: elements - 1; edges / neigh: x > 0 ? x - 1 : -1)
// Written against the facade's own containers (single pass over the file, no second read for the size as the reference does).
#ifndef FDAPDE_AMD_IO_H
#define FDAPDE_AMD_IO_H

#include <cstdlib>
#include <fstream>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#include "pde.h"

namespace fdapde {
namespace amd {

template <typename T> class CSVReader {
   public:
    CSVReader() = default;
    DMatrix<T> parse_file(const std::string& file) const {
        std::ifstream in(file, std::ios::binary);
        if (!in) throw std::runtime_error("CSVReader: cannot open " + file);
        std::string line, tok;
        if (!std::getline(in, line)) return DMatrix<T>();
        const int64_t cols = count_fields(line) - 1;   // the first column is the row index
        std::vector<T> values;
        int64_t rows = 0;
        while (std::getline(in, line)) {
            if (line.empty() || line == "\r") continue;
            int64_t field = 0;
            size_t pos = 0;
            while (pos <= line.size()) {
                const size_t end = std::min(line.find(',', pos), line.size());
                if (field > 0) {
                    if (field > cols) throw std::runtime_error("CSVReader: too many fields in a row of " + file);
                    tok.clear();
                    for (size_t k = pos; k < end; ++k) {
                        const char ch = line[k];
                        if (ch != ' ' && ch != '"' && ch != '\r') tok += ch;
                    }
                    values.push_back(convert(tok));
                }
                ++field, pos = end + 1;
            }
            if (field - 1 != cols) throw std::runtime_error("CSVReader: ragged row in " + file);
            ++rows;
        }
        DMatrix<T> m(rows, cols);
        for (int64_t i = 0; i < rows; ++i)
            for (int64_t j = 0; j < cols; ++j) m(i, j) = values[(size_t)(i * cols + j)];
        return m;
    }
   private:
    static int64_t count_fields(const std::string& line) {
        int64_t n = 1;
        for (char ch : line) n += ch == ',';
        return n;
    }
    static T convert(const std::string& tok) {
        if (tok == "NA" || tok == "NaN" || tok == "nan") return std::numeric_limits<T>::quiet_NaN();   // 0 for integral T, as in the reference
        if (tok.empty()) return T();
        return (T)std::strtod(tok.c_str(), nullptr);
    }
};

// loads <directory>/<mesh id>/{points, elements, boundary, edges, neigh}.csv; edges / neigh are optional (empty if absent)
template <int M, int N> struct MeshLoader {
    DMatrix<double> points_;
    DMatrix<int> elements_, edges_, boundary_, neighbors_;
    Triangulation<M, N> mesh;
    MeshLoader(const std::string& directory, const std::string& mesh_id) {
        const std::string base = directory + (directory.empty() || directory.back() == '/' ? "" : "/") + mesh_id + "/";
        CSVReader<double> dr;
        CSVReader<int> ir;
        points_ = dr.parse_file(base + "points.csv");
        elements_ = ir.parse_file(base + "elements.csv");
        for (int64_t i = 0; i < elements_.size(); ++i) elements_.data()[i] -= 1;   // realign indexes to 0
        boundary_ = ir.parse_file(base + "boundary.csv");
        auto optional = [&](const std::string& f) {
            std::ifstream probe(f);
            if (!probe) return DMatrix<int>();
            DMatrix<int> m = ir.parse_file(f);
            for (int64_t i = 0; i < m.size(); ++i) m.data()[i] = m.data()[i] > 0 ? m.data()[i] - 1 : -1;
            return m;
        };
        edges_ = optional(base + "edges.csv");
        neighbors_ = optional(base + "neigh.csv");
        mesh = Triangulation<M, N>(points_, elements_, boundary_);
    }
};

}   // namespace amd
}   // namespace fdapde

#endif
