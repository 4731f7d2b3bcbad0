# TMVBHip.jl -- ccall shim: TopicModelsVB.jl's `@gpu train!` on libtmvb_hip.so (MI355X / gfx950).
#
# Drop this file into src/ and `include("TMVBHip.jl")` after gpuCTPF.jl (src/TopicModelsVB.jl:28); `gpu_macro.jl`
# (same directory) replaces `macro gpu` (src/macros.jl:106-284).  The shim keeps the package's surface:
#   * hipLDA / hipCTM / hipCTPF <: TopicModel carry the fields that `@gpu`, `predict`, `topicdist`, `showtopics` read
#     (src/gpuLDA.jl:6-21, src/gpuCTM.jl:6-24, src/gpuCTPF.jl:6-46);
#   * one Julia function per device operator, as the reference has (src/gpuLDA.jl:132-340, src/gpuCTM.jl:166-480,
#     src/gpuCTPF.jl:314-670) -- the per-document operator chain of a sweep is ONE fused kernel here, so
#     update_phi!/update_gamma!/update_Elogtheta! (LDA), update_phi!/update_logzeta!/update_vsq!/update_lambda! (CTM) and
#     update_xi!/update_phi!/update_zayin!/update_gimel! (CTPF) are reached through `update_estep!`;
#   * train! has the signatures and defaults of src/gpuLDA.jl:347, src/gpuCTM.jl:487, src/gpuCTPF.jl:677; argument errors
#     are ArgumentError, state errors TopicModelError, corpus errors CorpusError, with the reference's messages;
#   * predict / topicdist methods for the hip types (src/modelutils.jl:831-913, :946-970);
#   * NEW: document-sharded multi-GPU train! behind the same call -- `hipComm` wraps the library's communicator (RCCL
#     over xGMI, or a host all-reduce callback), `train!(::Vector{hipLDA})` drives n GPUs from one host thread.
#
# NOTE: Julia is not installed in the build image of this repository, so this file has not been executed there.  Every
# ccall below is checked against include/tmvb.h by tests/test_julia_shim_static.py (name, arity, argument classes), and
# every call sequence is mirrored -- and tested on the GPU -- by the Python host in topicmodelsvb.jl_amd/*.py through
# the same C ABI.

const LIBTMVB = get(ENV, "TMVB_HIP_LIB", "libtmvb_hip.so")
# eight hardware queues for the process's HIP streams, before the runtime initialises (the library's constructor does the same: tmvb_core.hip)
haskey(ENV, "GPU_MAX_HW_QUEUES") || (ENV["GPU_MAX_HW_QUEUES"] = "8")

const TMVB_OK, TMVB_EINVAL, TMVB_ESHAPE, TMVB_ECORPUS, TMVB_ENOMEM, TMVB_EHIP, TMVB_ENONFINITE, TMVB_ENODEVICE, TMVB_ERCCL = 0, 1, 2, 3, 4, 5, 6, 7, 8
const TMVB_UNIQUE_ID_BYTES = 128

function tmvb_check(rc::Integer)
	rc == TMVB_OK && return nothing
	msg = unsafe_string(ccall((:tmvb_last_error, LIBTMVB), Cstring, ()))
	rc == TMVB_EINVAL     && throw(ArgumentError(msg))
	rc == TMVB_ESHAPE     && throw(TopicModelError(msg))
	rc == TMVB_ENONFINITE && throw(TopicModelError(msg))
	rc == TMVB_ECORPUS    && throw(CorpusError(msg))
	rc == TMVB_ENOMEM     && throw(OutOfMemoryError())
	error("libtmvb_hip (status $rc): " * msg)
end

# ---------------------------------------------------------------------------------------------- context and corpus

"cl.create_compute_context() (src/gpuLDA.jl:64): one GPU + one in-order stream."
function tmvb_context(device::Integer)
	ctx = Ref{Ptr{Cvoid}}(C_NULL)
	tmvb_check(ccall((:tmvb_ctx_create, LIBTMVB), Cint, (Int32, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), device, C_NULL, ctx))
	return ctx[]
end

"Corpus half of update_buffer! (src/modelutils.jl:370-388, :438-472): flat 0-based CSR."
function tmvb_upload_corpus(ctx::Ptr{Cvoid}, corp::Corpus)
	M, V, U = size(corp)
	doc_ptr = Int64[0; cumsum([length(doc.terms) for doc in corp])]
	terms   = Int32.(reduce(vcat, [doc.terms for doc in corp]; init=Int[]) .- 1)
	counts  = Int32.(reduce(vcat, [doc.counts for doc in corp]; init=Int[]))
	rdr_ptr = Int64[0; cumsum([length(doc.readers) for doc in corp])]
	readers = Int32.(reduce(vcat, [doc.readers for doc in corp]; init=Int[]) .- 1)
	ratings = Int32.(reduce(vcat, [doc.ratings for doc in corp]; init=Int[]))
	h = Ref{Ptr{Cvoid}}(C_NULL)
	GC.@preserve doc_ptr terms counts rdr_ptr readers ratings begin
		tmvb_check(ccall((:tmvb_corpus_create, LIBTMVB), Cint,
			(Ptr{Cvoid}, Int64, Int64, Int64, Ptr{Int64}, Ptr{Int32}, Ptr{Int32}, Ptr{Int64}, Ptr{Int32}, Ptr{Int32}, Ref{Ptr{Cvoid}}),
			ctx, M, V, U, doc_ptr, terms, counts, U > 0 ? pointer(rdr_ptr) : C_NULL,
			U > 0 ? pointer(readers) : C_NULL, U > 0 ? pointer(ratings) : C_NULL, h))
	end
	return h[]
end

tmvb_destroy_corpus(dcorp::Ptr{Cvoid}) = ccall((:tmvb_corpus_destroy, LIBTMVB), Cint, (Ptr{Cvoid},), dcorp)
tmvb_destroy_context(ctx::Ptr{Cvoid}) = ccall((:tmvb_ctx_destroy, LIBTMVB), Cint, (Ptr{Cvoid},), ctx)

# model.topics = [reverse(sortperm(vec(B[i,:]))) for i in 1:K] (src/gpuLDA.jl:374 and its siblings) as ONE stable segmented sort on the device,
# read backwards (tmvb_topic_order): K = 50 sorts of V = 25 319 values one after the other were the largest host item of a train! call.
function hip_topics(ctx::Ptr{Cvoid}, B::Matrix{Float64})
	K, V = size(B)
	out = Matrix{Int32}(undef, V, K)                 # the library's row-major [K][V] is this column-major V x K
	tmvb_check(ccall((:tmvb_topic_order, LIBTMVB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int32, Int64, Ptr{Int32}), ctx, B, K, V, out))
	[Int.(out[:, i]) for i in 1:K]
end

cols(m::Matrix{Float64}) = [m[:,d] for d in 1:size(m, 2)]
checkelbo_arg(checkelbo::Real) = checkelbo == Inf ? Int32(0) : Int32(checkelbo)

function check_train_args(tols, iters, checkelbo)
	all(tols .>= 0)														|| throw(ArgumentError("tolerance parameters must be nonnegative."))
	all(iters .>= 0)													|| throw(ArgumentError("iteration parameters must be nonnegative."))
	(isa(checkelbo, Integer) & (checkelbo > 0)) | (checkelbo == Inf)	|| throw(ArgumentError("checkelbo parameter must be a positive integer or Inf."))
end

"The printing half of check_elbo! (src/modelutils.jl:578-579); the first delta refers to the ELBO evaluated before the first iteration."
function print_delbo(traj::Vector{Float64}, done::Integer, baseline::Float64)
	prev = baseline
	for k in 1:done
		isnan(traj[k]) && continue
		println(k, " ∆elbo: ", round(traj[k] - prev, digits=3))
		prev = traj[k]
	end
end

# ---------------------------------------------------------------------------------------------- communicator (new)
# Document-sharded runs: every GPU holds a contiguous document shard; ONE all-reduce of the packed sufficient statistics
# per outer iteration happens inside the library (include/tmvb.h, "communicator").

mutable struct hipComm
	handle::Ptr{Cvoid}
	keep::Any                       # the @cfunction closure of a host transport must outlive the communicator
end

"128 bytes that rank 0 hands to every other rank (MPI.bcast, Distributed.jl, a file ...)."
function tmvb_unique_id()
	id = zeros(UInt8, TMVB_UNIQUE_ID_BYTES)
	tmvb_check(ccall((:tmvb_comm_unique_id, LIBTMVB), Cint, (Ptr{UInt8},), id))
	return id
end

"One process per GPU: ncclCommInitRank inside the library."
function hipComm(ctx::Ptr{Cvoid}, unique_id::Vector{UInt8}, nranks::Integer, rank::Integer)
	length(unique_id) == TMVB_UNIQUE_ID_BYTES || throw(ArgumentError("unique id must be $TMVB_UNIQUE_ID_BYTES bytes."))
	h = Ref{Ptr{Cvoid}}(C_NULL)
	tmvb_check(ccall((:tmvb_comm_create_rccl, LIBTMVB), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Int32, Int32, Ref{Ptr{Cvoid}}), ctx, unique_id, nranks, rank, h))
	return finalizer(c -> ccall((:tmvb_comm_destroy, LIBTMVB), Cint, (Ptr{Cvoid},), c.handle), hipComm(h[], nothing))
end

"One host thread, n GPUs: ncclCommInitAll over the contexts of the given models."
function hipComms(ctxs::Vector{Ptr{Cvoid}})
	out = fill(C_NULL, length(ctxs))
	tmvb_check(ccall((:tmvb_comm_create_rccl_all, LIBTMVB), Cint, (Ptr{Ptr{Cvoid}}, Int32, Ptr{Ptr{Cvoid}}), ctxs, length(ctxs), out))
	return [finalizer(c -> ccall((:tmvb_comm_destroy, LIBTMVB), Cint, (Ptr{Cvoid},), c.handle), hipComm(h, nothing)) for h in out]
end

"""
Host transport: `allreduce!(buf)` must leave the element-wise sum over all ranks in `buf` (a Vector{Float32} or
Vector{Float64} view of pinned host memory), e.g. `buf -> MPI.Allreduce!(buf, +, comm)`.
"""
function hipComm(ctx::Ptr{Cvoid}, allreduce!::Function, nranks::Integer, rank::Integer)
	function tramp(user::Ptr{Cvoid}, buf::Ptr{Cvoid}, count::Int64, dtype::Int32)::Cint
		try
			T = dtype == 0 ? Float32 : Float64
			allreduce!(unsafe_wrap(Array, Ptr{T}(buf), count))
			return Cint(0)
		catch
			return Cint(1)
		end
	end
	cb = @cfunction($tramp, Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int32))
	h = Ref{Ptr{Cvoid}}(C_NULL)
	tmvb_check(ccall((:tmvb_comm_create_host, LIBTMVB), Cint, (Ptr{Cvoid}, Int32, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), ctx, nranks, rank, cb, C_NULL, h))
	return finalizer(c -> ccall((:tmvb_comm_destroy, LIBTMVB), Cint, (Ptr{Cvoid},), c.handle), hipComm(h[], cb))
end

# ---------------------------------------------------------------------------------------------- LDA

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
	comm::Union{hipComm, Nothing}
end

"gpuLDA(corp, K) (src/gpuLDA.jl:45-84), built from the host model whose state it takes over (src/macros.jl:113-134)."
function hipLDA(model::LDA; device::Integer=0)
	ctx = tmvb_context(device)
	dcorp = tmvb_upload_corpus(ctx, model.corp)
	h = Ref{Ptr{Cvoid}}(C_NULL)
	tmvb_check(ccall((:tmvb_lda_create, LIBTMVB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ref{Ptr{Cvoid}}), ctx, dcorp, model.K, h))
	m = hipLDA(model.K, model.M, model.V, model.N, model.C, model.corp, model.topics, model.alpha, model.beta,
		model.beta_old, model.Elogtheta, model.Elogtheta_old, model.gamma, model.elbo, ctx, dcorp, h[], nothing)
	finalizer(m) do x
		ccall((:tmvb_lda_destroy, LIBTMVB), Cint, (Ptr{Cvoid},), x.handle)
		tmvb_destroy_corpus(x.dcorp)
		tmvb_destroy_context(x.ctx)
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
	model.gamma, model.Elogtheta, model.Elogtheta_old = cols(gamma), cols(El), cols(Elo)
	model.elbo = elbo[]
end

# one Julia function per device operator, as in src/gpuLDA.jl:132-340
"update_phi! / update_gamma! / update_Elogtheta! sweeps + update_beta!(model, d) of every document (src/LDA.jl:170-180)."
update_estep!(model::hipLDA, viter::Integer, vtol::Real) = tmvb_check(ccall((:tmvb_lda_estep, LIBTMVB), Cint, (Ptr{Cvoid}, Int32, Float64), model.handle, viter, vtol))
update_Elogtheta_sum!(model::hipLDA) = tmvb_check(ccall((:tmvb_lda_reduce_docs, LIBTMVB), Cint, (Ptr{Cvoid},), model.handle))
"Sharded run (set_comm!): update_estep! + update_Elogtheta_sum! + the statistics all-reduce, issued in vocabulary slabs under the last statistics pass."
update_estep_allreduce!(model::hipLDA, viter::Integer, vtol::Real) = tmvb_check(ccall((:tmvb_lda_estep_allreduce, LIBTMVB), Cint, (Ptr{Cvoid}, Int32, Float64), model.handle, viter, vtol))
update_beta!(model::hipLDA) = tmvb_check(ccall((:tmvb_lda_update_beta, LIBTMVB), Cint, (Ptr{Cvoid},), model.handle))
update_alpha!(model::hipLDA, niter::Integer, ntol::Real) = tmvb_check(ccall((:tmvb_lda_update_alpha, LIBTMVB), Cint, (Ptr{Cvoid}, Int32, Float64), model.handle, niter, ntol))
function update_elbo!(model::hipLDA)
	e = Ref{Float64}(0.0)
	tmvb_check(ccall((:tmvb_lda_update_elbo, LIBTMVB), Cint, (Ptr{Cvoid}, Ref{Float64}), model.handle, e))
	model.elbo = e[]
end

"Attach a communicator: this model's corpus is one document shard of a corpus of M_total documents."
function set_comm!(model::hipLDA, comm::Union{hipComm, Nothing}, M_total::Integer)
	tmvb_check(ccall((:tmvb_lda_set_comm, LIBTMVB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64), model.handle, comm === nothing ? C_NULL : comm.handle, M_total))
	model.comm = comm
	nothing
end

"""
    train!(model::hipLDA; iter=150, tol=1.0, niter=1000, ntol=1/K^2, viter=10, vtol=1/K^2, checkelbo=1, printelbo=true)

Same signature as train!(::gpuLDA) (src/gpuLDA.jl:347-376), semantics of the CPU path (src/LDA.jl:161-187: per-document
exit rule :175).  With a communicator attached every rank calls it with the same arguments (document-sharded train!).
"""
function train!(model::hipLDA; iter::Integer=150, tol::Real=1.0, niter::Integer=1000, ntol::Real=1/model.K^2, viter::Integer=10, vtol::Real=1/model.K^2, checkelbo::Real=1, printelbo::Bool=true)
	check_train_args([tol, ntol, vtol], [iter, niter, viter], checkelbo)
	update_buffer!(model)
	traj = fill(NaN, max(iter, 1)); done = Ref{Int32}(0); base = Ref{Float64}(model.elbo)
	tmvb_check(ccall((:tmvb_lda_train, LIBTMVB), Cint,
		(Ptr{Cvoid}, Int32, Float64, Int32, Float64, Int32, Float64, Int32, Ptr{Float64}, Ref{Int32}, Ref{Float64}),
		model.handle, iter, tol, niter, ntol, viter, vtol, checkelbo_arg(checkelbo), traj, done, base))
	printelbo && print_delbo(traj, done[], base[])
	(iter > 0) && update_host!(model)
	model.topics = hip_topics(model.ctx, model.beta)                             # reverse(sortperm(.)) per topic
	nothing
end

"""
    train!(models::Vector{hipLDA}; kwargs...)

One host thread, n GPUs: models[i] holds the i-th document shard on device i and the i-th communicator of
`hipComms([m.ctx for m in models])` (attach with set_comm!).  The n all-reduces of an iteration form one RCCL group.
"""
function train!(models::Vector{hipLDA}; iter::Integer=150, tol::Real=1.0, niter::Integer=1000, ntol::Real=1/models[1].K^2, viter::Integer=10, vtol::Real=1/models[1].K^2, checkelbo::Real=1, printelbo::Bool=true)
	check_train_args([tol, ntol, vtol], [iter, niter, viter], checkelbo)
	foreach(update_buffer!, models)
	hs = Ptr{Cvoid}[m.handle for m in models]
	traj = fill(NaN, max(iter, 1)); done = Ref{Int32}(0); base = Ref{Float64}(models[1].elbo)
	tmvb_check(ccall((:tmvb_lda_train_group, LIBTMVB), Cint,
		(Ptr{Ptr{Cvoid}}, Int32, Int32, Float64, Int32, Float64, Int32, Float64, Int32, Ptr{Float64}, Ref{Int32}, Ref{Float64}),
		hs, length(hs), iter, tol, niter, ntol, viter, vtol, checkelbo_arg(checkelbo), traj, done, base))
	printelbo && print_delbo(traj, done[], base[])
	for m in models
		(iter > 0) && update_host!(m)
		m.topics = hip_topics(m.ctx, m.beta)
	end
	nothing
end

"predict (src/modelutils.jl:831-855) on the device: the fused E-step with the trained alpha / beta frozen, no M-step."
function predict(corp::Corpus, train_model::hipLDA; iter::Integer=10, tol::Real=1/train_model.K^2)
	check_corp(corp)
	(corp.vocab == train_model.corp.vocab)	|| throw(CorpusError("predict corpus and train_model corpus must have identical vocabularies."))
	(tol >= 0)								|| throw(ArgumentError("tolerance parameter must be nonnegative."))
	(iter >= 0)								|| throw(ArgumentError("iteration parameter must be nonnegative."))
	host = LDA(corp, train_model.K)
	host.alpha, host.beta, host.beta_old, host.topics = train_model.alpha, train_model.beta, copy(train_model.beta), train_model.topics
	dev = hipLDA(host)
	update_buffer!(dev)
	update_estep!(dev, iter, tol)
	update_host!(dev)
	host.gamma, host.Elogtheta, host.Elogtheta_old = dev.gamma, dev.Elogtheta, dev.Elogtheta_old
	return host
end

function topicdist(model::hipLDA, d::Integer)       # src/modelutils.jl:946-951
	(d <= length(model.corp)) || throw(CorpusError("document index outside corpus range."))
	return model.gamma[d] / sum(model.gamma[d])
end

# ---------------------------------------------------------------------------------------------- CTM
# gpuCTM replacement (src/gpuCTM.jl:6-98).

mutable struct hipCTM <: TopicModel
	K::Int; M::Int; V::Int; N::Vector{Int}; C::Vector{Int}
	corp::Corpus; topics::VectorList{Int}
	mu::Vector{Float64}; sigma::Matrix{Float64}; invsigma::Matrix{Float64}
	beta::Matrix{Float64}; beta_old::Matrix{Float64}
	lambda::VectorList{Float64}; lambda_old::VectorList{Float64}; vsq::VectorList{Float64}; logzeta::Vector{Float64}
	elbo::Float64
	ctx::Ptr{Cvoid}; dcorp::Ptr{Cvoid}; handle::Ptr{Cvoid}
	comm::Union{hipComm, Nothing}
end

function hipCTM(model::CTM; device::Integer=0)
	ctx = tmvb_context(device)
	dcorp = tmvb_upload_corpus(ctx, model.corp)
	h = Ref{Ptr{Cvoid}}(C_NULL)
	tmvb_check(ccall((:tmvb_ctm_create, LIBTMVB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ref{Ptr{Cvoid}}), ctx, dcorp, model.K, h))
	m = hipCTM(model.K, model.M, model.V, model.N, model.C, model.corp, model.topics, model.mu, Matrix(model.sigma),
		Matrix(model.invsigma), model.beta, model.beta_old, model.lambda, model.lambda_old, model.vsq, model.logzeta,
		model.elbo, ctx, dcorp, h[], nothing)
	finalizer(m) do x
		ccall((:tmvb_ctm_destroy, LIBTMVB), Cint, (Ptr{Cvoid},), x.handle)
		tmvb_destroy_corpus(x.dcorp)
		tmvb_destroy_context(x.ctx)
	end
	return m
end

"update_buffer! (src/modelutils.jl:400-436): beta_old / lambda_old travel too (update_elbo! rebuilds phi from them, src/CTM.jl:93)."
function update_buffer!(model::hipCTM)
	beta, beta_old = Matrix{Float64}(model.beta), Matrix{Float64}(model.beta_old)
	lam, lam_old, vsq = hcat(model.lambda...), hcat(model.lambda_old...), hcat(model.vsq...)
	elbo = Ref{Float64}(model.elbo)
	GC.@preserve beta beta_old lam lam_old vsq begin
		tmvb_check(ccall((:tmvb_ctm_set_state, LIBTMVB), Cint,
			(Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Float64}),
			model.handle, model.mu, model.sigma, model.invsigma, beta, beta_old, lam, lam_old, vsq, model.logzeta, elbo))
	end
end

function update_host!(model::hipCTM)        # src/modelutils.jl:519-537
	K, M, V = model.K, model.M, model.V
	mu = zeros(K); sg = zeros(K, K); isg = zeros(K, K); beta = zeros(K, V); beta_old = zeros(K, V)
	lam = zeros(K, M); lam_old = zeros(K, M); vsq = zeros(K, M); lz = zeros(M); elbo = Ref{Float64}(0.0)
	tmvb_check(ccall((:tmvb_ctm_get_state, LIBTMVB), Cint,
		(Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Float64}),
		model.handle, mu, sg, isg, beta, beta_old, lam, lam_old, vsq, lz, elbo))
	model.mu, model.sigma, model.invsigma, model.beta, model.beta_old = mu, sg, isg, beta, beta_old
	model.lambda, model.lambda_old, model.vsq = cols(lam), cols(lam_old), cols(vsq)
	model.logzeta = lz; model.elbo = elbo[]
end

# one Julia function per device operator (src/gpuCTM.jl:166-480)
"update_phi! / update_logzeta! / update_vsq! / update_lambda! sweeps + update_beta!(model, d) of every document (src/CTM.jl:194-205)."
update_estep!(model::hipCTM, niter::Integer, ntol::Real, viter::Integer, vtol::Real) = tmvb_check(ccall((:tmvb_ctm_estep, LIBTMVB), Cint, (Ptr{Cvoid}, Int32, Float64, Int32, Float64), model.handle, niter, ntol, viter, vtol))
"sum_d lambda_d, sum_d vsq_d and the MFMA scatter matrix with the current (= previous) mu."
update_doc_sums!(model::hipCTM) = tmvb_check(ccall((:tmvb_ctm_reduce_docs, LIBTMVB), Cint, (Ptr{Cvoid},), model.handle))
update_beta!(model::hipCTM) = tmvb_check(ccall((:tmvb_ctm_update_beta, LIBTMVB), Cint, (Ptr{Cvoid},), model.handle))      # src/gpuCTM.jl:253
update_sigma!(model::hipCTM) = tmvb_check(ccall((:tmvb_ctm_update_sigma, LIBTMVB), Cint, (Ptr{Cvoid},), model.handle))    # :200 -- before update_mu! (quirk Q2)
update_mu!(model::hipCTM) = tmvb_check(ccall((:tmvb_ctm_update_mu, LIBTMVB), Cint, (Ptr{Cvoid},), model.handle))          # :166
function update_elbo!(model::hipCTM)
	e = Ref{Float64}(0.0)
	tmvb_check(ccall((:tmvb_ctm_update_elbo, LIBTMVB), Cint, (Ptr{Cvoid}, Ref{Float64}), model.handle, e))
	model.elbo = e[]
end

function set_comm!(model::hipCTM, comm::Union{hipComm, Nothing}, M_total::Integer)
	tmvb_check(ccall((:tmvb_ctm_set_comm, LIBTMVB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64), model.handle, comm === nothing ? C_NULL : comm.handle, M_total))
	model.comm = comm
	nothing
end

"""
    train!(model::hipCTM; iter=150, tol=1.0, niter=1000, ntol=1/K^2, viter=10, vtol=1/K^2, checkelbo=1, printelbo=true)

Signature of train!(::gpuCTM) (src/gpuCTM.jl:487-519), semantics of the CPU path (src/CTM.jl:185-213).
"""
function train!(model::hipCTM; iter::Integer=150, tol::Real=1.0, niter::Integer=1000, ntol::Real=1/model.K^2, viter::Integer=10, vtol::Real=1/model.K^2, checkelbo::Real=1, printelbo::Bool=true)
	check_train_args([tol, ntol, vtol], [iter, niter, viter], checkelbo)
	update_buffer!(model)
	traj = fill(NaN, max(iter, 1)); done = Ref{Int32}(0); base = Ref{Float64}(model.elbo)
	tmvb_check(ccall((:tmvb_ctm_train, LIBTMVB), Cint,
		(Ptr{Cvoid}, Int32, Float64, Int32, Float64, Int32, Float64, Int32, Ptr{Float64}, Ref{Int32}, Ref{Float64}),
		model.handle, iter, tol, niter, ntol, viter, vtol, checkelbo_arg(checkelbo), traj, done, base))
	printelbo && print_delbo(traj, done[], base[])
	(iter > 0) && update_host!(model)
	model.topics = hip_topics(model.ctx, model.beta)                             # reverse(sortperm(.)) per topic
	nothing
end

function train!(models::Vector{hipCTM}; iter::Integer=150, tol::Real=1.0, niter::Integer=1000, ntol::Real=1/models[1].K^2, viter::Integer=10, vtol::Real=1/models[1].K^2, checkelbo::Real=1, printelbo::Bool=true)
	check_train_args([tol, ntol, vtol], [iter, niter, viter], checkelbo)
	foreach(update_buffer!, models)
	hs = Ptr{Cvoid}[m.handle for m in models]
	traj = fill(NaN, max(iter, 1)); done = Ref{Int32}(0); base = Ref{Float64}(models[1].elbo)
	tmvb_check(ccall((:tmvb_ctm_train_group, LIBTMVB), Cint,
		(Ptr{Ptr{Cvoid}}, Int32, Int32, Float64, Int32, Float64, Int32, Float64, Int32, Ptr{Float64}, Ref{Int32}, Ref{Float64}),
		hs, length(hs), iter, tol, niter, ntol, viter, vtol, checkelbo_arg(checkelbo), traj, done, base))
	printelbo && print_delbo(traj, done[], base[])
	for m in models
		(iter > 0) && update_host!(m)
		m.topics = hip_topics(m.ctx, m.beta)
	end
	nothing
end

"predict (src/modelutils.jl:886-913) on the device: the fused CTM E-step with mu / sigma / beta frozen."
function predict(corp::Corpus, train_model::hipCTM; iter::Integer=10, tol::Real=1/train_model.K^2, niter::Integer=1000, ntol::Real=1/train_model.K^2)
	check_corp(corp)
	(corp.vocab == train_model.corp.vocab)	|| throw(CorpusError("predict corpus and train_model corpus must have identical vocabularies."))
	all([tol, ntol] .>= 0)					|| throw(ArgumentError("tolerance parameters must be nonnegative."))
	all([iter, niter] .>= 0)				|| throw(ArgumentError("iteration parameters must be nonnegative."))
	host = CTM(corp, train_model.K)
	host.mu, host.sigma, host.invsigma = train_model.mu, Symmetric(train_model.sigma), Symmetric(train_model.invsigma)
	host.beta, host.beta_old, host.topics = train_model.beta, copy(train_model.beta), train_model.topics
	dev = hipCTM(host)
	update_buffer!(dev)
	update_estep!(dev, niter, ntol, iter, tol)
	update_host!(dev)
	host.lambda, host.lambda_old, host.vsq, host.logzeta = dev.lambda, dev.lambda_old, dev.vsq, dev.logzeta
	return host
end

function topicdist(model::hipCTM, d::Integer)       # src/modelutils.jl:953-958
	(d <= length(model.corp)) || throw(CorpusError("document index outside corpus range."))
	return additive_logistic(model.lambda[d] + 0.5 * model.vsq[d])
end

# ---------------------------------------------------------------------------------------------- CTPF
# gpuCTPF replacement (src/gpuCTPF.jl:6-153).  train! ends with the recommendation tail of src/gpuCTPF.jl:711-731,
# which here is ONE call (device GEMM + segmented sorts) instead of M + U host sortperm calls.

mutable struct hipCTPF <: TopicModel
	K::Int; M::Int; V::Int; U::Int
	N::Vector{Int}; C::Vector{Int}; R::Vector{Int}
	corp::Corpus; topics::VectorList{Int}
	scores::Matrix{Float64}; libs::VectorList{Int}; drecs::VectorList{Int}; urecs::VectorList{Int}
	hyper::Vector{Float64}                                   # a, b, c, d, e, f, g, h  (src/CTPF.jl:81)
	alef::Matrix{Float64}; alef_old::Matrix{Float64}; he::Matrix{Float64}; he_old::Matrix{Float64}
	bet::Vector{Float64}; bet_old::Vector{Float64}; vav::Vector{Float64}; vav_old::Vector{Float64}
	dalet::Vector{Float64}; dalet_old::Vector{Float64}; het::Vector{Float64}; het_old::Vector{Float64}
	gimel::VectorList{Float64}; gimel_old::VectorList{Float64}; zayin::VectorList{Float64}; zayin_old::VectorList{Float64}
	elbo::Float64
	ctx::Ptr{Cvoid}; dcorp::Ptr{Cvoid}; handle::Ptr{Cvoid}
	comm::Union{hipComm, Nothing}
end

function hipCTPF(model::CTPF; device::Integer=0)
	ctx = tmvb_context(device)
	dcorp = tmvb_upload_corpus(ctx, model.corp)
	h = Ref{Ptr{Cvoid}}(C_NULL)
	tmvb_check(ccall((:tmvb_ctpf_create, LIBTMVB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ref{Ptr{Cvoid}}), ctx, dcorp, model.K, h))
	m = hipCTPF(model.K, model.M, model.V, model.U, model.N, model.C, model.R, model.corp, model.topics, model.scores, model.libs,
		model.drecs, model.urecs, Float64[model.a, model.b, model.c, model.d, model.e, model.f, model.g, model.h],
		model.alef, model.alef_old, model.he, model.he_old, model.bet, model.bet_old, model.vav, model.vav_old,
		model.dalet, model.dalet_old, model.het, model.het_old, model.gimel, model.gimel_old, model.zayin, model.zayin_old,
		model.elbo, ctx, dcorp, h[], nothing)
	finalizer(m) do x
		ccall((:tmvb_ctpf_destroy, LIBTMVB), Cint, (Ptr{Cvoid},), x.handle)
		tmvb_destroy_corpus(x.dcorp)
		tmvb_destroy_context(x.ctx)
	end
	return m
end

"update_buffer! (src/modelutils.jl:438-494), the *_old fields included (update_elbo! rebuilds phi / xi from them, src/CTPF.jl:239-240)."
function update_buffer!(model::hipCTPF)
	gim, zay = hcat(model.gimel...), hcat(model.zayin...)
	gim_old, zay_old = hcat(model.gimel_old...), hcat(model.zayin_old...)
	elbo = Ref{Float64}(model.elbo)
	GC.@preserve gim zay gim_old zay_old begin
		tmvb_check(ccall((:tmvb_ctpf_set_state, LIBTMVB), Cint,
			(Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Float64}),
			model.handle, model.hyper, model.alef, model.he, model.bet, model.vav, model.dalet, model.het, gim, zay, elbo))
		tmvb_check(ccall((:tmvb_ctpf_set_state_old, LIBTMVB), Cint,
			(Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
			model.handle, model.alef_old, model.he_old, model.bet_old, model.vav_old, model.dalet_old, model.het_old, gim_old, zay_old))
	end
end

function update_host!(model::hipCTPF)       # src/modelutils.jl:539-570
	K, M, V, U = model.K, model.M, model.V, model.U
	alef = zeros(K, V); alef_old = zeros(K, V); he = zeros(K, U); he_old = zeros(K, U); rates = zeros(8K)
	gim = zeros(K, M); gim_old = zeros(K, M); zay = zeros(K, M); zay_old = zeros(K, M); elbo = Ref{Float64}(0.0)
	tmvb_check(ccall((:tmvb_ctpf_get_state, LIBTMVB), Cint,
		(Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Float64}),
		model.handle, alef, alef_old, he, he_old, rates, gim, gim_old, zay, zay_old, elbo))
	model.alef, model.alef_old, model.he, model.he_old = alef, alef_old, he, he_old
	model.bet, model.vav, model.dalet, model.het = rates[1:K], rates[K+1:2K], rates[2K+1:3K], rates[3K+1:4K]
	model.bet_old, model.vav_old, model.dalet_old, model.het_old = rates[4K+1:5K], rates[5K+1:6K], rates[6K+1:7K], rates[7K+1:8K]
	model.gimel, model.gimel_old, model.zayin, model.zayin_old = cols(gim), cols(gim_old), cols(zay), cols(zay_old)
	model.elbo = elbo[]
end

# one Julia function per device operator (src/gpuCTPF.jl:314-670)
"update_xi! / update_phi! / update_zayin! / update_gimel! sweeps + update_he!(d) / update_alef!(d) of every document (src/CTPF.jl:353-365)."
update_estep!(model::hipCTPF, viter::Integer, vtol::Real) = tmvb_check(ccall((:tmvb_ctpf_estep, LIBTMVB), Cint, (Ptr{Cvoid}, Int32, Float64), model.handle, viter, vtol))
"sum_d gimel_d, sum_d zayin_d (inputs of update_bet! / update_vav!)."
update_doc_sums!(model::hipCTPF) = tmvb_check(ccall((:tmvb_ctpf_reduce_docs, LIBTMVB), Cint, (Ptr{Cvoid},), model.handle))
"update_he!, update_alef!, update_dalet!, update_het!, update_bet!, update_vav! in the reference's order (src/CTPF.jl:366-371)."
update_globals!(model::hipCTPF) = tmvb_check(ccall((:tmvb_ctpf_mstep, LIBTMVB), Cint, (Ptr{Cvoid},), model.handle))
function update_elbo!(model::hipCTPF)
	e = Ref{Float64}(0.0)
	tmvb_check(ccall((:tmvb_ctpf_update_elbo, LIBTMVB), Cint, (Ptr{Cvoid}, Ref{Float64}), model.handle, e))
	model.elbo = e[]
end

function set_comm!(model::hipCTPF, comm::Union{hipComm, Nothing})
	tmvb_check(ccall((:tmvb_ctpf_set_comm, LIBTMVB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), model.handle, comm === nothing ? C_NULL : comm.handle))
	model.comm = comm
	nothing
end

"scores / drecs / urecs of src/CTPF.jl:379-399 from the device-resident state (ids converted to 1-based)."
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
	check_train_args([tol, vtol], [iter, viter], checkelbo)
	update_buffer!(model)
	traj = fill(NaN, max(iter, 1)); done = Ref{Int32}(0); base = Ref{Float64}(model.elbo)
	tmvb_check(ccall((:tmvb_ctpf_train, LIBTMVB), Cint,
		(Ptr{Cvoid}, Int32, Float64, Int32, Float64, Int32, Ptr{Float64}, Ref{Int32}, Ref{Float64}),
		model.handle, iter, tol, viter, vtol, checkelbo_arg(checkelbo), traj, done, base))
	printelbo && print_delbo(traj, done[], base[])
	(iter > 0) && update_host!(model)
	Ebeta = model.alef ./ model.bet
	model.topics = hip_topics(model.ctx, Matrix{Float64}(Ebeta))                # reverse(sortperm(.)) per topic, src/CTPF.jl:376-377
	(model.comm === nothing) && update_recs!(model)                             # :379-399 (a shard ranks only its own documents)
	nothing
end

function topicdist(model::hipCTPF, d::Integer)      # src/modelutils.jl:960-965
	(d <= length(model.corp)) || throw(CorpusError("document index outside corpus range."))
	return model.gimel[d] / sum(model.gimel[d])
end

# ---------------------------------------------------------------------------------------------- fLDA / fCTM
# Filtered models (src/fLDA.jl, src/fCTM.jl).  The reference has no device types for them: `@gpu train!` on an fLDA / fCTM
# does nothing (src/macros.jl:274-278).  hipfLDA / hipfCTM follow the gpuLDA / gpuCTM pattern on the engine's filtered
# kernels.  tau / tau_old travel as flat vectors in corpus token order (vcat(model.tau...)).

splitdocs(flat::Vector{Float64}, N::Vector{Int}) = (o = cumsum([0; N]); [flat[o[d]+1:o[d+1]] for d in 1:length(N)])
flatdocs(v::VectorList{Float64}) = isempty(v) ? Float64[] : vcat(v...)

mutable struct hipfLDA <: TopicModel
	K::Int; M::Int; V::Int; N::Vector{Int}; C::Vector{Int}
	corp::Corpus; topics::VectorList{Int}
	eta::Float64; alpha::Vector{Float64}
	kappa::Vector{Float64}; kappa_old::Vector{Float64}
	beta::Matrix{Float64}; beta_old::Matrix{Float64}
	Elogtheta::VectorList{Float64}; Elogtheta_old::VectorList{Float64}; gamma::VectorList{Float64}
	tau::VectorList{Float64}; tau_old::VectorList{Float64}
	elbo::Float64
	ctx::Ptr{Cvoid}; dcorp::Ptr{Cvoid}; handle::Ptr{Cvoid}
	comm::Union{hipComm, Nothing}
end

function hipfLDA(model::fLDA; device::Integer=0)
	ctx = tmvb_context(device)
	dcorp = tmvb_upload_corpus(ctx, model.corp)
	h = Ref{Ptr{Cvoid}}(C_NULL)
	tmvb_check(ccall((:tmvb_flda_create, LIBTMVB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ref{Ptr{Cvoid}}), ctx, dcorp, model.K, h))
	m = hipfLDA(model.K, model.M, model.V, model.N, model.C, model.corp, model.topics, model.eta, model.alpha, model.kappa, model.kappa_old,
		model.beta, model.beta_old, model.Elogtheta, model.Elogtheta_old, model.gamma, model.tau, model.tau_old, model.elbo, ctx, dcorp, h[], nothing)
	finalizer(m) do x
		ccall((:tmvb_flda_destroy, LIBTMVB), Cint, (Ptr{Cvoid},), x.handle)
		tmvb_destroy_corpus(x.dcorp)
		tmvb_destroy_context(x.ctx)
	end
	return m
end

function update_buffer!(model::hipfLDA)
	beta, beta_old = Matrix{Float64}(model.beta), Matrix{Float64}(model.beta_old)
	gamma, El, Elo = hcat(model.gamma...), hcat(model.Elogtheta...), hcat(model.Elogtheta_old...)
	tau, tau_old = flatdocs(model.tau), flatdocs(model.tau_old)
	eta = Ref{Float64}(model.eta); elbo = Ref{Float64}(model.elbo)
	GC.@preserve beta beta_old gamma El Elo tau tau_old begin
		tmvb_check(ccall((:tmvb_flda_set_state, LIBTMVB), Cint,
			(Ptr{Cvoid}, Ref{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Float64}),
			model.handle, eta, model.alpha, model.kappa, model.kappa_old, beta, beta_old, gamma, El, Elo, tau, tau_old, elbo))
	end
end

function update_host!(model::hipfLDA)
	K, M, V, nnz = model.K, model.M, model.V, sum(model.N)
	alpha = zeros(K); kappa = zeros(V); kappa_old = zeros(V); beta = zeros(K, V); beta_old = zeros(K, V)
	gamma = zeros(K, M); El = zeros(K, M); Elo = zeros(K, M); tau = zeros(nnz); tau_old = zeros(nnz)
	eta = Ref{Float64}(0.0); elbo = Ref{Float64}(0.0)
	tmvb_check(ccall((:tmvb_flda_get_state, LIBTMVB), Cint,
		(Ptr{Cvoid}, Ref{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Float64}),
		model.handle, eta, alpha, kappa, kappa_old, beta, beta_old, gamma, El, Elo, tau, tau_old, elbo))
	model.eta, model.alpha, model.kappa, model.kappa_old, model.beta, model.beta_old = eta[], alpha, kappa, kappa_old, beta, beta_old
	model.gamma, model.Elogtheta, model.Elogtheta_old = cols(gamma), cols(El), cols(Elo)
	model.tau, model.tau_old = splitdocs(tau, model.N), splitdocs(tau_old, model.N)
	model.elbo = elbo[]
end

"update_phi! / update_tau! / update_gamma! / update_Elogtheta! sweeps + update_beta!(model, d) + update_kappa!(model, d) (src/fLDA.jl:222-236)."
update_estep!(model::hipfLDA, viter::Integer, vtol::Real) = tmvb_check(ccall((:tmvb_flda_estep, LIBTMVB), Cint, (Ptr{Cvoid}, Int32, Float64), model.handle, viter, vtol))
update_Elogtheta_sum!(model::hipfLDA) = tmvb_check(ccall((:tmvb_flda_reduce_docs, LIBTMVB), Cint, (Ptr{Cvoid},), model.handle))
"update_beta! and update_kappa! (src/fLDA.jl:152, :138)."
update_beta!(model::hipfLDA) = tmvb_check(ccall((:tmvb_flda_update_beta, LIBTMVB), Cint, (Ptr{Cvoid},), model.handle))
update_alpha!(model::hipfLDA, niter::Integer, ntol::Real) = tmvb_check(ccall((:tmvb_flda_update_alpha, LIBTMVB), Cint, (Ptr{Cvoid}, Int32, Float64), model.handle, niter, ntol))
update_eta!(model::hipfLDA) = tmvb_check(ccall((:tmvb_flda_update_eta, LIBTMVB), Cint, (Ptr{Cvoid},), model.handle))      # src/fLDA.jl:122
function update_elbo!(model::hipfLDA)
	e = Ref{Float64}(0.0)
	tmvb_check(ccall((:tmvb_flda_update_elbo, LIBTMVB), Cint, (Ptr{Cvoid}, Ref{Float64}), model.handle, e))
	model.elbo = e[]
end

"Document shard of a corpus of M_total documents and C_total tokens (sum of all counts; eta's denominator, src/fLDA.jl:123)."
function set_comm!(model::hipfLDA, comm::Union{hipComm, Nothing}, M_total::Integer, C_total::Integer)
	tmvb_check(ccall((:tmvb_flda_set_comm, LIBTMVB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64, Int64), model.handle, comm === nothing ? C_NULL : comm.handle, M_total, C_total))
	model.comm = comm
	nothing
end

"train!(model::fLDA; ...) (src/fLDA.jl:213-247) on the device."
function train!(model::hipfLDA; iter::Integer=150, tol::Real=1.0, niter::Integer=1000, ntol::Real=1/model.K^2, viter::Integer=10, vtol::Real=1/model.K^2, checkelbo::Real=1, printelbo::Bool=true)
	check_train_args([tol, ntol, vtol], [iter, niter, viter], checkelbo)
	update_buffer!(model)
	traj = fill(NaN, max(iter, 1)); done = Ref{Int32}(0); base = Ref{Float64}(model.elbo)
	tmvb_check(ccall((:tmvb_flda_train, LIBTMVB), Cint,
		(Ptr{Cvoid}, Int32, Float64, Int32, Float64, Int32, Float64, Int32, Ptr{Float64}, Ref{Int32}, Ref{Float64}),
		model.handle, iter, tol, niter, ntol, viter, vtol, checkelbo_arg(checkelbo), traj, done, base))
	printelbo && print_delbo(traj, done[], base[])
	(iter > 0) && update_host!(model)
	model.topics = hip_topics(model.ctx, model.beta)                             # reverse(sortperm(.)) per topic
	nothing
end

function train!(models::Vector{hipfLDA}; iter::Integer=150, tol::Real=1.0, niter::Integer=1000, ntol::Real=1/models[1].K^2, viter::Integer=10, vtol::Real=1/models[1].K^2, checkelbo::Real=1, printelbo::Bool=true)
	check_train_args([tol, ntol, vtol], [iter, niter, viter], checkelbo)
	foreach(update_buffer!, models)
	hs = Ptr{Cvoid}[m.handle for m in models]
	traj = fill(NaN, max(iter, 1)); done = Ref{Int32}(0); base = Ref{Float64}(models[1].elbo)
	tmvb_check(ccall((:tmvb_flda_train_group, LIBTMVB), Cint,
		(Ptr{Ptr{Cvoid}}, Int32, Int32, Float64, Int32, Float64, Int32, Float64, Int32, Ptr{Float64}, Ref{Int32}, Ref{Float64}),
		hs, length(hs), iter, tol, niter, ntol, viter, vtol, checkelbo_arg(checkelbo), traj, done, base))
	printelbo && print_delbo(traj, done[], base[])
	for m in models
		(iter > 0) && update_host!(m)
		m.topics = hip_topics(m.ctx, m.beta)
	end
	nothing
end

"""
predict (src/modelutils.jl:857-883) on the device.  As in the reference only alpha, beta and topics come from the trained model
(:866-868): kappa and eta are those of the fresh fLDA(corp, K).  The reference's loop tests `vtol`, which that function never
defines (:877); `tol` is used here.
"""
function predict(corp::Corpus, train_model::hipfLDA; iter::Integer=10, tol::Real=1/train_model.K^2)
	check_corp(corp)
	(corp.vocab == train_model.corp.vocab)	|| throw(CorpusError("predict corpus and train_model corpus must have identical vocabularies."))
	(tol >= 0)								|| throw(ArgumentError("tolerance parameter must be nonnegative."))
	(iter >= 0)								|| throw(ArgumentError("iteration parameter must be nonnegative."))
	host = fLDA(corp, train_model.K)
	host.alpha, host.beta, host.beta_old, host.topics = train_model.alpha, train_model.beta, copy(train_model.beta), train_model.topics
	dev = hipfLDA(host)
	update_buffer!(dev)
	update_estep!(dev, iter, tol)
	update_host!(dev)
	host.gamma, host.Elogtheta, host.Elogtheta_old, host.tau, host.tau_old = dev.gamma, dev.Elogtheta, dev.Elogtheta_old, dev.tau, dev.tau_old
	return host
end

function topicdist(model::hipfLDA, d::Integer)      # src/modelutils.jl:946-951
	(d <= length(model.corp)) || throw(CorpusError("document index outside corpus range."))
	return model.gamma[d] / sum(model.gamma[d])
end

mutable struct hipfCTM <: TopicModel
	K::Int; M::Int; V::Int; N::Vector{Int}; C::Vector{Int}
	corp::Corpus; topics::VectorList{Int}
	eta::Float64
	mu::Vector{Float64}; sigma::Matrix{Float64}; invsigma::Matrix{Float64}
	kappa::Vector{Float64}; kappa_old::Vector{Float64}
	beta::Matrix{Float64}; beta_old::Matrix{Float64}
	lambda::VectorList{Float64}; lambda_old::VectorList{Float64}; vsq::VectorList{Float64}; logzeta::Vector{Float64}
	tau::VectorList{Float64}; tau_old::VectorList{Float64}
	elbo::Float64
	ctx::Ptr{Cvoid}; dcorp::Ptr{Cvoid}; handle::Ptr{Cvoid}
	comm::Union{hipComm, Nothing}
end

function hipfCTM(model::fCTM; device::Integer=0)
	ctx = tmvb_context(device)
	dcorp = tmvb_upload_corpus(ctx, model.corp)
	h = Ref{Ptr{Cvoid}}(C_NULL)
	tmvb_check(ccall((:tmvb_fctm_create, LIBTMVB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ref{Ptr{Cvoid}}), ctx, dcorp, model.K, h))
	m = hipfCTM(model.K, model.M, model.V, model.N, model.C, model.corp, model.topics, model.eta, model.mu, Matrix(model.sigma),
		Matrix(model.invsigma), model.kappa, model.kappa_old, model.beta, model.beta_old, model.lambda, model.lambda_old, model.vsq,
		model.logzeta, model.tau, model.tau_old, model.elbo, ctx, dcorp, h[], nothing)
	finalizer(m) do x
		ccall((:tmvb_fctm_destroy, LIBTMVB), Cint, (Ptr{Cvoid},), x.handle)
		tmvb_destroy_corpus(x.dcorp)
		tmvb_destroy_context(x.ctx)
	end
	return m
end

function update_buffer!(model::hipfCTM)
	beta, beta_old = Matrix{Float64}(model.beta), Matrix{Float64}(model.beta_old)
	lam, lam_old, vsq = hcat(model.lambda...), hcat(model.lambda_old...), hcat(model.vsq...)
	tau, tau_old = flatdocs(model.tau), flatdocs(model.tau_old)
	eta = Ref{Float64}(model.eta); elbo = Ref{Float64}(model.elbo)
	GC.@preserve beta beta_old lam lam_old vsq tau tau_old begin
		tmvb_check(ccall((:tmvb_fctm_set_state, LIBTMVB), Cint,
			(Ptr{Cvoid}, Ref{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Float64}),
			model.handle, eta, model.mu, model.sigma, model.invsigma, model.kappa, model.kappa_old, beta, beta_old, lam, lam_old, vsq, model.logzeta, tau, tau_old, elbo))
	end
end

function update_host!(model::hipfCTM)
	K, M, V, nnz = model.K, model.M, model.V, sum(model.N)
	mu = zeros(K); sg = zeros(K, K); isg = zeros(K, K); kappa = zeros(V); kappa_old = zeros(V); beta = zeros(K, V); beta_old = zeros(K, V)
	lam = zeros(K, M); lam_old = zeros(K, M); vsq = zeros(K, M); lz = zeros(M); tau = zeros(nnz); tau_old = zeros(nnz)
	eta = Ref{Float64}(0.0); elbo = Ref{Float64}(0.0)
	tmvb_check(ccall((:tmvb_fctm_get_state, LIBTMVB), Cint,
		(Ptr{Cvoid}, Ref{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ref{Float64}),
		model.handle, eta, mu, sg, isg, kappa, kappa_old, beta, beta_old, lam, lam_old, vsq, lz, tau, tau_old, elbo))
	model.eta, model.mu, model.sigma, model.invsigma = eta[], mu, sg, isg
	model.kappa, model.kappa_old, model.beta, model.beta_old = kappa, kappa_old, beta, beta_old
	model.lambda, model.lambda_old, model.vsq, model.logzeta = cols(lam), cols(lam_old), cols(vsq), lz
	model.tau, model.tau_old = splitdocs(tau, model.N), splitdocs(tau_old, model.N)
	model.elbo = elbo[]
end

"update_phi! / update_tau! / update_logzeta! / update_lambda! / update_vsq! sweeps + update_beta!(model, d) + update_kappa!(model, d) (src/fCTM.jl:233-248)."
update_estep!(model::hipfCTM, niter::Integer, ntol::Real, viter::Integer, vtol::Real) = tmvb_check(ccall((:tmvb_fctm_estep, LIBTMVB), Cint, (Ptr{Cvoid}, Int32, Float64, Int32, Float64), model.handle, niter, ntol, viter, vtol))
update_doc_sums!(model::hipfCTM) = tmvb_check(ccall((:tmvb_fctm_reduce_docs, LIBTMVB), Cint, (Ptr{Cvoid},), model.handle))
"update_beta! and update_kappa! (src/fCTM.jl:148, :134)."
update_beta!(model::hipfCTM) = tmvb_check(ccall((:tmvb_fctm_update_beta, LIBTMVB), Cint, (Ptr{Cvoid},), model.handle))
update_sigma!(model::hipfCTM) = tmvb_check(ccall((:tmvb_fctm_update_sigma, LIBTMVB), Cint, (Ptr{Cvoid},), model.handle))  # src/fCTM.jl:128 -- before update_mu!
update_mu!(model::hipfCTM) = tmvb_check(ccall((:tmvb_fctm_update_mu, LIBTMVB), Cint, (Ptr{Cvoid},), model.handle))        # :122
function update_elbo!(model::hipfCTM)
	e = Ref{Float64}(0.0)
	tmvb_check(ccall((:tmvb_fctm_update_elbo, LIBTMVB), Cint, (Ptr{Cvoid}, Ref{Float64}), model.handle, e))
	model.elbo = e[]
end

function set_comm!(model::hipfCTM, comm::Union{hipComm, Nothing}, M_total::Integer)
	tmvb_check(ccall((:tmvb_fctm_set_comm, LIBTMVB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Int64), model.handle, comm === nothing ? C_NULL : comm.handle, M_total))
	model.comm = comm
	nothing
end

"train!(model::fCTM; ...) (src/fCTM.jl:226-262) on the device; eta stays fixed (update_eta! is commented out, :253)."
function train!(model::hipfCTM; iter::Integer=150, tol::Real=1.0, niter::Integer=1000, ntol::Real=1/model.K^2, viter::Integer=10, vtol::Real=1/model.K^2, checkelbo::Real=1, printelbo::Bool=true)
	check_train_args([tol, ntol, vtol], [iter, niter, viter], checkelbo)
	update_buffer!(model)
	traj = fill(NaN, max(iter, 1)); done = Ref{Int32}(0); base = Ref{Float64}(model.elbo)
	tmvb_check(ccall((:tmvb_fctm_train, LIBTMVB), Cint,
		(Ptr{Cvoid}, Int32, Float64, Int32, Float64, Int32, Float64, Int32, Ptr{Float64}, Ref{Int32}, Ref{Float64}),
		model.handle, iter, tol, niter, ntol, viter, vtol, checkelbo_arg(checkelbo), traj, done, base))
	printelbo && print_delbo(traj, done[], base[])
	(iter > 0) && update_host!(model)
	model.topics = hip_topics(model.ctx, model.beta)                             # reverse(sortperm(.)) per topic
	nothing
end

function train!(models::Vector{hipfCTM}; iter::Integer=150, tol::Real=1.0, niter::Integer=1000, ntol::Real=1/models[1].K^2, viter::Integer=10, vtol::Real=1/models[1].K^2, checkelbo::Real=1, printelbo::Bool=true)
	check_train_args([tol, ntol, vtol], [iter, niter, viter], checkelbo)
	foreach(update_buffer!, models)
	hs = Ptr{Cvoid}[m.handle for m in models]
	traj = fill(NaN, max(iter, 1)); done = Ref{Int32}(0); base = Ref{Float64}(models[1].elbo)
	tmvb_check(ccall((:tmvb_fctm_train_group, LIBTMVB), Cint,
		(Ptr{Ptr{Cvoid}}, Int32, Int32, Float64, Int32, Float64, Int32, Float64, Int32, Ptr{Float64}, Ref{Int32}, Ref{Float64}),
		hs, length(hs), iter, tol, niter, ntol, viter, vtol, checkelbo_arg(checkelbo), traj, done, base))
	printelbo && print_delbo(traj, done[], base[])
	for m in models
		(iter > 0) && update_host!(m)
		m.topics = hip_topics(m.ctx, m.beta)
	end
	nothing
end

"predict (src/modelutils.jl:915-943) on the device; mu / sigma / invsigma / beta / topics from the trained model (:924-928), `tol` for the reference's undefined `vtol` (:937)."
function predict(corp::Corpus, train_model::hipfCTM; iter::Integer=10, tol::Real=1/train_model.K^2, niter::Integer=1000, ntol::Real=1/train_model.K^2)
	check_corp(corp)
	(corp.vocab == train_model.corp.vocab)	|| throw(CorpusError("predict corpus and train_model corpus must have identical vocabularies."))
	all([tol, ntol] .>= 0)					|| throw(ArgumentError("tolerance parameters must be nonnegative."))
	all([iter, niter] .>= 0)				|| throw(ArgumentError("iteration parameters must be nonnegative."))
	host = fCTM(corp, train_model.K)
	host.mu, host.sigma, host.invsigma = train_model.mu, Symmetric(train_model.sigma), Symmetric(train_model.invsigma)
	host.beta, host.beta_old, host.topics = train_model.beta, copy(train_model.beta), train_model.topics
	dev = hipfCTM(host)
	update_buffer!(dev)
	update_estep!(dev, niter, ntol, iter, tol)
	update_host!(dev)
	host.lambda, host.lambda_old, host.vsq, host.logzeta, host.tau, host.tau_old = dev.lambda, dev.lambda_old, dev.vsq, dev.logzeta, dev.tau, dev.tau_old
	return host
end

function topicdist(model::hipfCTM, d::Integer)      # src/modelutils.jl:953-958
	(d <= length(model.corp)) || throw(CorpusError("document index outside corpus range."))
	return additive_logistic(model.lambda[d] + 0.5 * model.vsq[d])
end
