# TMVBHip.jl -- ccall shim: TopicModelsVB.jl's `@gpu train!` on libtmvb_hip.so (MI355X / gfx950).
#
# Drop this file into src/ and `include("TMVBHip.jl")` after gpuCTPF.jl (src/TopicModelsVB.jl:28).
# It keeps the package's surface: `hipLDA <: TopicModel` has the fields of gpuLDA that `@gpu`,
# `predict`, `topicdist`, `showtopics` read (src/gpuLDA.jl:6-21), `train!(::hipLDA; ...)` has the
# signature and defaults of src/gpuLDA.jl:347, argument errors are ArgumentError, state errors
# TopicModelError, corpus errors CorpusError.
#
# NOTE: Julia is not installed in the build image of this repository, so this file has not been
# executed there; every call it makes is mirrored (and tested on the GPU) by the Python host in
# topicmodelsvb.jl_amd/lda.py through the same C ABI (include/tmvb.h).

const LIBTMVB = get(ENV, "TMVB_HIP_LIB", "libtmvb_hip.so")

const TMVB_OK, TMVB_EINVAL, TMVB_ESHAPE, TMVB_ECORPUS = 0, 1, 2, 3

function tmvb_check(rc::Integer)
	rc == TMVB_OK && return nothing
	msg = unsafe_string(ccall((:tmvb_last_error, LIBTMVB), Cstring, ()))
	rc == TMVB_EINVAL  && throw(ArgumentError(msg))
	rc == TMVB_ESHAPE  && throw(TopicModelError(msg))
	rc == TMVB_ECORPUS && throw(CorpusError(msg))
	rc == 6            && throw(TopicModelError(msg))   # TMVB_ENONFINITE
	error("libtmvb_hip: " * msg)
end

mutable struct hipLDA <: TopicModel
	K::Int
	M::Int
	V::Int
	N::Vector{Int}
	C::Vector{Int}
	corp::Corpus
	topics::VectorList{Int}
	alpha::Vector{Float64}
	beta::Matrix{Float64}
	beta_old::Matrix{Float64}
	Elogtheta::VectorList{Float64}
	Elogtheta_old::VectorList{Float64}
	gamma::VectorList{Float64}
	elbo::Float64
	ctx::Ptr{Cvoid}
	dcorp::Ptr{Cvoid}
	handle::Ptr{Cvoid}
end

"Corpus half of update_buffer! (src/modelutils.jl:370-388): flat 0-based CSR."
function tmvb_upload_corpus(ctx::Ptr{Cvoid}, corp::Corpus)
	M, V, U = size(corp)
	doc_ptr = Int64[0; cumsum([length(doc.terms) for doc in corp])]
	terms   = Int32.(vcat([doc.terms for doc in corp]...) .- 1)
	counts  = Int32.(vcat([doc.counts for doc in corp]...))
	rdr_ptr = Int64[0; cumsum([length(doc.readers) for doc in corp])]
	readers = Int32.(vcat([doc.readers for doc in corp]...) .- 1)
	ratings = Int32.(vcat([doc.ratings for doc in corp]...))
	h = Ref{Ptr{Cvoid}}(C_NULL)
	GC.@preserve doc_ptr terms counts rdr_ptr readers ratings begin
		tmvb_check(ccall((:tmvb_corpus_create, LIBTMVB), Cint,
			(Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Int64}, Ptr{Int32}, Ptr{Int32}, Ptr{Int64}, Ptr{Int32}, Ptr{Int32}, Ref{Ptr{Cvoid}}),
			ctx, M, V, U, doc_ptr, terms, counts, U > 0 ? pointer(rdr_ptr) : C_NULL,
			U > 0 ? pointer(readers) : C_NULL, U > 0 ? pointer(ratings) : C_NULL, h))
	end
	return h[]
end

function hipLDA(model::LDA; device::Integer=0)
	ctx = Ref{Ptr{Cvoid}}(C_NULL)
	tmvb_check(ccall((:tmvb_ctx_create, LIBTMVB), Cint, (Int32, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), device, C_NULL, ctx))
	dcorp = tmvb_upload_corpus(ctx[], model.corp)
	h = Ref{Ptr{Cvoid}}(C_NULL)
	tmvb_check(ccall((:tmvb_lda_create, LIBTMVB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ref{Ptr{Cvoid}}), ctx[], dcorp, model.K, h))
	m = hipLDA(model.K, model.M, model.V, model.N, model.C, model.corp, model.topics, model.alpha, model.beta,
		model.beta_old, model.Elogtheta, model.Elogtheta_old, model.gamma, model.elbo, ctx[], dcorp, h[])
	finalizer(m) do x
		ccall((:tmvb_lda_destroy, LIBTMVB), Cint, (Ptr{Cvoid},), x.handle)
		ccall((:tmvb_corpus_destroy, LIBTMVB), Cint, (Ptr{Cvoid},), x.dcorp)
		ccall((:tmvb_ctx_destroy, LIBTMVB), Cint, (Ptr{Cvoid},), x.ctx)
	end
	return m
end

"State half of update_buffer! (src/modelutils.jl:390-396)."
function update_buffer!(model::hipLDA)
	beta, beta_old = Matrix{Float64}(model.beta), Matrix{Float64}(model.beta_old)   # column-major K x V
	gamma, El, Elo = hcat(model.gamma...), hcat(model.Elogtheta...), hcat(model.Elogtheta_old...)
	elbo = Ref{Float64}(model.elbo)
	GC.@preserve beta beta_old gamma El Elo begin
		tmvb_check(ccall((:tmvb_lda_set_state, LIBTMVB), Cint,
			(Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Float64}),
			model.handle, model.alpha, beta, beta_old, gamma, El, Elo, elbo))
	end
end

"update_host! (src/modelutils.jl:501-516); phi is never materialised."
function update_host!(model::hipLDA)
	K, M, V = model.K, model.M, model.V
	alpha = zeros(K); beta = zeros(K, V); beta_old = zeros(K, V)
	gamma = zeros(K, M); El = zeros(K, M); Elo = zeros(K, M); elbo = Ref{Float64}(0.0)
	tmvb_check(ccall((:tmvb_lda_get_state, LIBTMVB), Cint,
		(Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Float64}),
		model.handle, alpha, beta, beta_old, gamma, El, Elo, elbo))
	model.alpha, model.beta, model.beta_old = alpha, beta, beta_old
	model.gamma = [gamma[:,d] for d in 1:M]
	model.Elogtheta = [El[:,d] for d in 1:M]
	model.Elogtheta_old = [Elo[:,d] for d in 1:M]
	model.elbo = elbo[]
end

# one Julia function per device operator, as in src/gpuLDA.jl:132-340
update_estep!(model::hipLDA, viter::Integer, vtol::Real) = tmvb_check(ccall((:tmvb_lda_estep, LIBTMVB), Cint, (Ptr{Cvoid}, Int32, Float64), model.handle, viter, vtol))
update_Elogtheta_sum!(model::hipLDA) = tmvb_check(ccall((:tmvb_lda_reduce_docs, LIBTMVB), Cint, (Ptr{Cvoid},), model.handle))
update_beta!(model::hipLDA) = tmvb_check(ccall((:tmvb_lda_update_beta, LIBTMVB), Cint, (Ptr{Cvoid},), model.handle))
update_alpha!(model::hipLDA, niter::Integer, ntol::Real) = tmvb_check(ccall((:tmvb_lda_update_alpha, LIBTMVB), Cint, (Ptr{Cvoid}, Int32, Float64), model.handle, niter, ntol))
function update_elbo!(model::hipLDA)
	e = Ref{Float64}(0.0)
	tmvb_check(ccall((:tmvb_lda_update_elbo, LIBTMVB), Cint, (Ptr{Cvoid}, Ref{Float64}), model.handle, e))
	model.elbo = e[]
end

"""
    train!(model::hipLDA; iter=150, tol=1.0, niter=1000, ntol=1/K^2, viter=10, vtol=1/K^2, checkelbo=1, printelbo=true)

Same signature and semantics as train!(::gpuLDA) (src/gpuLDA.jl:347-376) with the CPU path's
per-document exit rule (src/LDA.jl:175).
"""
function train!(model::hipLDA; iter::Integer=150, tol::Real=1.0, niter::Integer=1000, ntol::Real=1/model.K^2, viter::Integer=10, vtol::Real=1/model.K^2, checkelbo::Real=1, printelbo::Bool=true)
	all([tol, ntol, vtol] .>= 0)										|| throw(ArgumentError("tolerance parameters must be nonnegative."))
	all([iter, niter, viter] .>= 0)										|| throw(ArgumentError("iteration parameters must be nonnegative."))
	(isa(checkelbo, Integer) & (checkelbo > 0)) | (checkelbo == Inf)	|| throw(ArgumentError("checkelbo parameter must be a positive integer or Inf."))
	update_buffer!(model)
	traj = fill(NaN, max(iter, 1)); done = Ref{Int32}(0)
	tmvb_check(ccall((:tmvb_lda_train, LIBTMVB), Cint,
		(Ptr{Cvoid}, Int32, Float64, Int32, Float64, Int32, Float64, Int32, Ptr{Float64}, Ref{Int32}),
		model.handle, iter, tol, niter, ntol, viter, vtol, checkelbo == Inf ? 0 : Int(checkelbo), traj, done))
	if printelbo
		prev = model.elbo
		for k in 1:done[]
			isnan(traj[k]) && continue
			println(k, " ∆elbo: ", round(traj[k] - prev, digits=3)); prev = traj[k]
		end
	end
	(iter > 0) && update_host!(model)
	model.topics = [reverse(sortperm(vec(model.beta[i,:]))) for i in 1:model.K]
	nothing
end

# ---------------------------------------------------------------------------------------------- CTM
# gpuCTM replacement (src/gpuCTM.jl:6-98).  Same pattern as hipLDA: the handle owns the device state, the Julia
# fields are the host copies that update_buffer!/update_host! move.

mutable struct hipCTM <: TopicModel
	K::Int; M::Int; V::Int; N::Vector{Int}; C::Vector{Int}
	corp::Corpus; topics::VectorList{Int}
	mu::Vector{Float64}; sigma::Matrix{Float64}; invsigma::Matrix{Float64}
	beta::Matrix{Float64}; beta_old::Matrix{Float64}
	lambda::VectorList{Float64}; lambda_old::VectorList{Float64}; vsq::VectorList{Float64}; logzeta::Vector{Float64}
	elbo::Float64
	ctx::Ptr{Cvoid}; dcorp::Ptr{Cvoid}; handle::Ptr{Cvoid}
end

function hipCTM(model::CTM; device::Integer=0)
	ctx = Ref{Ptr{Cvoid}}(C_NULL); h = Ref{Ptr{Cvoid}}(C_NULL)
	tmvb_check(ccall((:tmvb_ctx_create, LIBTMVB), Cint, (Int32, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), device, C_NULL, ctx))
	dcorp = tmvb_upload_corpus(ctx[], model.corp)
	tmvb_check(ccall((:tmvb_ctm_create, LIBTMVB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ref{Ptr{Cvoid}}), ctx[], dcorp, model.K, h))
	m = hipCTM(model.K, model.M, model.V, model.N, model.C, model.corp, model.topics, model.mu, Matrix(model.sigma),
		Matrix(model.invsigma), model.beta, copy(model.beta), model.lambda, deepcopy(model.lambda), model.vsq, model.logzeta,
		model.elbo, ctx[], dcorp, h[])
	finalizer(m) do x
		ccall((:tmvb_ctm_destroy, LIBTMVB), Cint, (Ptr{Cvoid},), x.handle)
		ccall((:tmvb_corpus_destroy, LIBTMVB), Cint, (Ptr{Cvoid},), x.dcorp)
		ccall((:tmvb_ctx_destroy, LIBTMVB), Cint, (Ptr{Cvoid},), x.ctx)
	end
	m
end

function update_buffer!(model::hipCTM)      # src/modelutils.jl:400-436
	lam = hcat(model.lambda...); vsq = hcat(model.vsq...); elbo = Ref(model.elbo)
	GC.@preserve model lam vsq tmvb_check(ccall((:tmvb_ctm_set_state, LIBTMVB), Cint,
		(Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Float64}),
		model.handle, model.mu, model.sigma, model.invsigma, model.beta, C_NULL, lam, C_NULL, vsq, model.logzeta, elbo))
end

function update_host!(model::hipCTM)        # src/modelutils.jl:519-537
	K, M, V = model.K, model.M, model.V
	mu = zeros(K); sg = zeros(K, K); isg = zeros(K, K); beta = zeros(K, V); beta_old = zeros(K, V)
	lam = zeros(K, M); lam_old = zeros(K, M); vsq = zeros(K, M); lz = zeros(M); elbo = Ref(0.0)
	tmvb_check(ccall((:tmvb_ctm_get_state, LIBTMVB), Cint,
		(Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Float64}),
		model.handle, mu, sg, isg, beta, beta_old, lam, lam_old, vsq, lz, elbo))
	model.mu, model.sigma, model.invsigma, model.beta, model.beta_old = mu, sg, isg, beta, beta_old
	model.lambda = [lam[:,d] for d in 1:M]; model.lambda_old = [lam_old[:,d] for d in 1:M]
	model.vsq = [vsq[:,d] for d in 1:M]; model.logzeta = lz; model.elbo = elbo[]
end

"""
    train!(model::hipCTM; iter=150, tol=1.0, niter=1000, ntol=1/K^2, viter=10, vtol=1/K^2, checkelbo=1, printelbo=true)

Signature of train!(::gpuCTM) (src/gpuCTM.jl:487-519), semantics of the CPU path (src/CTM.jl:185-213).
"""
function train!(model::hipCTM; iter::Integer=150, tol::Real=1.0, niter::Integer=1000, ntol::Real=1/model.K^2, viter::Integer=10, vtol::Real=1/model.K^2, checkelbo::Real=1, printelbo::Bool=true)
	all([tol, ntol, vtol] .>= 0)										|| throw(ArgumentError("tolerance parameters must be nonnegative."))
	all([iter, niter, viter] .>= 0)										|| throw(ArgumentError("iteration parameters must be nonnegative."))
	(isa(checkelbo, Integer) & (checkelbo > 0)) | (checkelbo == Inf)	|| throw(ArgumentError("checkelbo parameter must be a positive integer or Inf."))
	update_buffer!(model)
	traj = fill(NaN, max(iter, 1)); done = Ref{Int32}(0)
	tmvb_check(ccall((:tmvb_ctm_train, LIBTMVB), Cint,
		(Ptr{Cvoid}, Int32, Float64, Int32, Float64, Int32, Float64, Int32, Ptr{Float64}, Ref{Int32}),
		model.handle, iter, tol, niter, ntol, viter, vtol, checkelbo == Inf ? 0 : Int(checkelbo), traj, done))
	(iter > 0) && update_host!(model)
	model.topics = [reverse(sortperm(vec(model.beta[i,:]))) for i in 1:model.K]
	nothing
end

# ---------------------------------------------------------------------------------------------- CTPF
# gpuCTPF replacement (src/gpuCTPF.jl:6-153).  train! ends with the recommendation tail of src/gpuCTPF.jl:711-731,
# which here is ONE call (device GEMM + segmented sorts) instead of M + U host sortperm calls.

mutable struct hipCTPF <: TopicModel
	K::Int; M::Int; V::Int; U::Int
	corp::Corpus; topics::VectorList{Int}
	scores::Matrix{Float64}; libs::VectorList{Int}; drecs::VectorList{Int}; urecs::VectorList{Int}
	hyper::Vector{Float64}                                   # a, b, c, d, e, f, g, h  (src/CTPF.jl:81)
	alef::Matrix{Float64}; he::Matrix{Float64}
	bet::Vector{Float64}; vav::Vector{Float64}; dalet::Vector{Float64}; het::Vector{Float64}
	gimel::VectorList{Float64}; zayin::VectorList{Float64}
	elbo::Float64
	ctx::Ptr{Cvoid}; dcorp::Ptr{Cvoid}; handle::Ptr{Cvoid}
end

function hipCTPF(model::CTPF; device::Integer=0)
	ctx = Ref{Ptr{Cvoid}}(C_NULL); h = Ref{Ptr{Cvoid}}(C_NULL)
	tmvb_check(ccall((:tmvb_ctx_create, LIBTMVB), Cint, (Int32, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), device, C_NULL, ctx))
	dcorp = tmvb_upload_corpus(ctx[], model.corp)
	tmvb_check(ccall((:tmvb_ctpf_create, LIBTMVB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ref{Ptr{Cvoid}}), ctx[], dcorp, model.K, h))
	m = hipCTPF(model.K, model.M, model.V, model.U, model.corp, model.topics, model.scores, model.libs, model.drecs, model.urecs,
		Float64[model.a, model.b, model.c, model.d, model.e, model.f, model.g, model.h], model.alef, model.he,
		model.bet, model.vav, model.dalet, model.het, model.gimel, model.zayin, model.elbo, ctx[], dcorp, h[])
	finalizer(m) do x
		ccall((:tmvb_ctpf_destroy, LIBTMVB), Cint, (Ptr{Cvoid},), x.handle)
		ccall((:tmvb_corpus_destroy, LIBTMVB), Cint, (Ptr{Cvoid},), x.dcorp)
		ccall((:tmvb_ctx_destroy, LIBTMVB), Cint, (Ptr{Cvoid},), x.ctx)
	end
	m
end

function update_buffer!(model::hipCTPF)     # src/modelutils.jl:438-494
	gim = hcat(model.gimel...); zay = hcat(model.zayin...); elbo = Ref(model.elbo)
	GC.@preserve model gim zay tmvb_check(ccall((:tmvb_ctpf_set_state, LIBTMVB), Cint,
		(Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Float64}),
		model.handle, model.hyper, model.alef, model.he, model.bet, model.vav, model.dalet, model.het, gim, zay, elbo))
end

function update_host!(model::hipCTPF)       # src/modelutils.jl:539-570
	K, M, V, U = model.K, model.M, model.V, model.U
	alef = zeros(K, V); he = zeros(K, U); rates = zeros(8K); gim = zeros(K, M); zay = zeros(K, M); elbo = Ref(0.0)
	tmvb_check(ccall((:tmvb_ctpf_get_state, LIBTMVB), Cint,
		(Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Float64}),
		model.handle, alef, C_NULL, he, C_NULL, rates, gim, C_NULL, zay, C_NULL, elbo))
	model.alef, model.he = alef, he
	model.bet, model.vav, model.dalet, model.het = rates[1:K], rates[K+1:2K], rates[2K+1:3K], rates[3K+1:4K]
	model.gimel = [gim[:,d] for d in 1:M]; model.zayin = [zay[:,d] for d in 1:M]; model.elbo = elbo[]
end

# scores / drecs / urecs of src/CTPF.jl:379-399 from the device-resident state (ids converted to 1-based)
function update_recs!(model::hipCTPF)
	M, U = model.M, model.U
	scores = zeros(M, U); dr = zeros(Int32, U, M); dc = zeros(Int32, M); ur = zeros(Int32, M, U); uc = zeros(Int32, U)
	# the C arrays are row-major [M][U] / [U][M]: a column-major Julia (U, M) / (M, U) matrix has the same memory layout
	tmvb_check(ccall((:tmvb_ctpf_recommend, LIBTMVB), Cint,
		(Ptr{Cvoid}, Ptr{Float64}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Ptr{Cfloat}, Ptr{Cfloat}),
		model.handle, scores, dr, dc, ur, uc, C_NULL, C_NULL))
	model.scores = scores
	model.drecs = [Int.(dr[1:dc[d], d]) .+ 1 for d in 1:M]
	model.urecs = [Int.(ur[1:uc[u], u]) .+ 1 for u in 1:U]
	nothing
end

"""
    train!(model::hipCTPF; iter=150, tol=1.0, viter=10, vtol=1/K^2, checkelbo=Inf, printelbo=true)

Signature of train!(::gpuCTPF) (src/gpuCTPF.jl:677-733), semantics of the CPU path (src/CTPF.jl:344-400).
"""
function train!(model::hipCTPF; iter::Integer=150, tol::Real=1.0, viter::Integer=10, vtol::Real=1/model.K^2, checkelbo::Real=Inf, printelbo::Bool=true)
	all([tol, vtol] .>= 0)												|| throw(ArgumentError("tolerance parameters must be nonnegative."))
	all([iter, viter] .>= 0)											|| throw(ArgumentError("iteration parameters must be nonnegative."))
	(isa(checkelbo, Integer) & (checkelbo > 0)) | (checkelbo == Inf)	|| throw(ArgumentError("checkelbo parameter must be a positive integer or Inf."))
	update_buffer!(model)
	traj = fill(NaN, max(iter, 1)); done = Ref{Int32}(0)
	tmvb_check(ccall((:tmvb_ctpf_train, LIBTMVB), Cint,
		(Ptr{Cvoid}, Int32, Float64, Int32, Float64, Int32, Ptr{Float64}, Ref{Int32}),
		model.handle, iter, tol, viter, vtol, checkelbo == Inf ? 0 : Int(checkelbo), traj, done))
	(iter > 0) && update_host!(model)
	Ebeta = model.alef ./ model.bet
	model.topics = [reverse(sortperm(vec(Ebeta[i,:]))) for i in 1:model.K]      # src/CTPF.jl:376-377
	update_recs!(model)                                                         # :379-399
	nothing
end

# Inside `macro gpu` (src/macros.jl:113-150) the LDA branch becomes:
#
#     if isa(model, LDA)
#         gpumodel = hipLDA(model)
#         train!(gpumodel; kwargs...)
#         model.topics, model.alpha, model.beta = gpumodel.topics, gpumodel.alpha, gpumodel.beta
#         model.Elogtheta = gpumodel.Elogtheta;  model.Elogtheta_old = deepcopy(model.Elogtheta)
#         model.gamma, model.elbo = gpumodel.gamma, gpumodel.elbo
#         model.beta ./= sum(model.beta, dims=2);  model.beta_old = copy(model.beta)      # :147-148
#         nothing
