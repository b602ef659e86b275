# npy_io.jl — the .npy subset the upstream scripts exchange with NumPy (oracle_dump.jl, reference_cpu_baseline.jl)
# ---- minimal .npy (v1.0, little-endian Float64, C order) reader / writer -------------------------------------------------
function read_npy(path)
    open(path) do io
        read(io, 6) == UInt8[0x93, 'N', 'U', 'M', 'P', 'Y'] || error("not an .npy file: $path")
        read(io, 2); hlen = Int(read(io, UInt16))
        header = String(read(io, hlen))
        occursin("'<f8'", header) && occursin("'fortran_order': False", header) || error("expected C-order <f8: $header")
        dims = parse.(Int, split(strip(match(r"\(([^)]*)\)", header).captures[1], [' ', ',']), r"\s*,\s*"; keepempty = false))
        data = Vector{Float64}(undef, prod(dims)); read!(io, data)
        # 1-D: the vector; 2-D: rows = j, columns = i; N-D: NumPy's index order (C order ⇒ reverse the dims, then reverse the axes)
        length(dims) == 1 ? data : permutedims(reshape(data, reverse(dims)...), length(dims):-1:1)
    end
end

function write_npy(path, A::AbstractMatrix{Float64})   # A[j, i] → C-order (ny, nx)
    header = "{'descr': '<f8', 'fortran_order': False, 'shape': ($(size(A, 1)), $(size(A, 2))), }"
    header *= " "^(63 - (10 + length(header)) % 64) * "\n"
    open(path, "w") do io
        write(io, UInt8[0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0]); write(io, UInt16(length(header))); write(io, header)
        write(io, collect(permutedims(A)))
    end
end

