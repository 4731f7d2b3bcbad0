# gpu_macro.jl -- `macro gpu` on the HIP engine: replaces src/macros.jl:106-284.
#
# The reference's macro builds an empty gpuLDA / gpuCTM / gpuCTPF, assigns every field of the host model to it,
# rebuilds phi (and xi) on the host from the *_old fields, trains, and copies the fields back.  The hip types are
# constructed FROM the host model (they take its fields over, src/macros.jl:115-134 / :153-175 / :196-229) and never
# materialise phi / xi -- the device rebuilds the last-sweep responsibilities from the *_old state inside update_elbo!
# (src/LDA.jl:87-88, src/CTM.jl:93, src/CTPF.jl:239-240) -- so each branch is: construct, train!, copy back.
# model.phi (and xi) are rebuilt on the host from the DEVICE's *_old state (dev_phi1 / dev_xi1 below).
# The copy-back keeps the reference's post-conditions: *_old = copy of the new value, beta re-normalised in Float64
# (src/macros.jl:147-148, :188-189) so that check_model's isprobvec holds for the Float32-derived rows.
#
# (Julia is not installed in this repository's build image; see TMVBHip.jl.)

# model.phi / model.xi after `@gpu train!`: the reference hands back gpumodel.phi[1] (and xi[1]) -- the responsibilities of the
# LAST sweep the device ran (src/macros.jl:144, :186, :259-260).  The engine never materialises them; it keeps what they are
# made of: after train! the device's *_old state (beta_old = the beta the last E-step read, Elogtheta_old / lambda_old /
# gimel_old ... = the input of each document's last sweep; filled into `dev` by update_host!).  So they are rebuilt from
# dev.*_old -- never from the host model's pre-training *_old fields.

"Last-sweep phi of the first document from the device's state (update_phi!, src/LDA.jl:150-154)."
function dev_phi1(dev::hipLDA)
	dev.M == 0 && return Matrix{Float64}[]
	phi = EPSILON .+ dev.beta_old[:, dev.corp[1].terms] .* exp.(dev.Elogtheta_old[1])
	return [phi ./ sum(phi, dims=1)]
end

"Last-sweep phi of the first document from the device's state (update_phi!, src/CTM.jl:175-178)."
function dev_phi1(dev::hipCTM)
	dev.M == 0 && return Matrix{Float64}[]
	return [additive_logistic(log.(dev.beta_old[:, dev.corp[1].terms]) .+ dev.lambda_old[1], dims=1)]
end

"Last-sweep phi of the first document from the device's state (update_phi!, src/CTPF.jl:327-330)."
function dev_phi1(dev::hipCTPF)
	dev.M == 0 && return Matrix{Float64}[]
	phi = exp.(digamma.(dev.gimel_old[1]) - log.(dev.dalet_old) - log.(dev.bet_old) .+ digamma.(dev.alef_old[:, dev.corp[1].terms]))
	return [phi ./ sum(phi, dims=1)]
end

"Last-sweep xi of the first document from the device's state (update_xi!, src/CTPF.jl:334-337)."
function dev_xi1(dev::hipCTPF)
	dev.M == 0 && return Matrix{Float64}[]
	psi_he = digamma.(dev.he_old[:, dev.corp[1].readers])
	xi = vcat(exp.(digamma.(dev.gimel_old[1]) - log.(dev.dalet_old) - log.(dev.vav_old) .+ psi_he),
	          exp.(digamma.(dev.zayin_old[1]) - log.(dev.het_old) - log.(dev.vav_old) .+ psi_he))
	return [xi ./ sum(xi, dims=1)]
end

"Last-sweep phi of the first document of a filtered model from the device's state (src/fLDA.jl:204-207)."
function dev_phi1(dev::hipfLDA)
	dev.M == 0 && return Matrix{Float64}[]
	terms = dev.corp[1].terms
	return [additive_logistic(dev.tau_old[1]' .* log.(@boink dev.beta_old[:,terms]) .+ dev.Elogtheta_old[1], dims=1)]
end

"Last-sweep phi of the first document of a filtered model from the device's state (src/fCTM.jl:230-233)."
function dev_phi1(dev::hipfCTM)
	dev.M == 0 && return Matrix{Float64}[]
	terms = dev.corp[1].terms
	return [additive_logistic(dev.tau_old[1]' .* log.(@boink dev.beta_old[:,terms]) .+ dev.lambda_old[1], dims=1)]
end

function copyback!(model::LDA, dev::hipLDA)                        # src/macros.jl:136-149
	model.topics = dev.topics
	model.alpha = dev.alpha
	model.beta = dev.beta
	model.Elogtheta = dev.Elogtheta
	model.Elogtheta_old = deepcopy(model.Elogtheta)                # :140
	model.gamma = dev.gamma
	model.phi = dev_phi1(dev)                                      # :142 (gpumodel.phi[1]: the device's last sweep)
	model.elbo = dev.elbo
	model.beta ./= sum(model.beta, dims=2)                         # :145
	model.beta_old = copy(model.beta)                              # :146
	nothing
end

function copyback!(model::CTM, dev::hipCTM)                        # src/macros.jl:177-192
	model.topics = dev.topics
	model.mu = dev.mu
	model.sigma = Symmetric(dev.sigma)                             # :179
	model.invsigma = Symmetric(dev.invsigma)                       # :180
	model.beta = dev.beta
	model.lambda = dev.lambda
	model.lambda_old = deepcopy(model.lambda)                      # :183
	model.vsq = dev.vsq
	model.logzeta = dev.logzeta
	model.phi = dev_phi1(dev)                                      # :186
	model.elbo = dev.elbo
	model.beta ./= sum(model.beta, dims=2)                         # :189
	model.beta_old = copy(model.beta)                              # :190
	nothing
end

function copyback!(model::CTPF, dev::hipCTPF)                      # src/macros.jl:239-265
	model.topics = dev.topics
	model.scores = dev.scores
	model.drecs = dev.drecs
	model.urecs = dev.urecs
	model.phi = dev_phi1(dev)                                      # :259, from the device's *_old state, before it is overwritten below
	model.xi = dev_xi1(dev)                                        # :260
	model.alef = dev.alef
	model.alef_old = copy(model.alef)
	model.he = dev.he
	model.he_old = copy(model.he)
	model.bet = dev.bet
	model.bet_old = copy(model.bet)
	model.vav = dev.vav
	model.vav_old = copy(model.vav)
	model.gimel = dev.gimel
	model.gimel_old = deepcopy(model.gimel)
	model.zayin = dev.zayin
	model.zayin_old = deepcopy(model.zayin)
	model.dalet = dev.dalet
	model.dalet_old = copy(model.dalet)
	model.het = dev.het
	model.het_old = copy(model.het)
	model.elbo = dev.elbo
	nothing
end

function copyback!(model::fLDA, dev::hipfLDA)
	model.topics = dev.topics
	model.eta = dev.eta
	model.alpha = dev.alpha
	model.kappa = dev.kappa
	model.beta = dev.beta
	model.Elogtheta = dev.Elogtheta
	model.Elogtheta_old = deepcopy(model.Elogtheta)
	model.gamma = dev.gamma
	model.phi = dev_phi1(dev)
	model.tau = dev.tau
	model.tau_old = deepcopy(model.tau)
	model.elbo = dev.elbo
	model.beta ./= sum(model.beta, dims=2); model.beta_old = copy(model.beta)          # Float64 re-normalisation as src/macros.jl:147-148
	model.kappa ./= sum(model.kappa); model.kappa_old = copy(model.kappa)
	nothing
end

function copyback!(model::fCTM, dev::hipfCTM)
	model.topics = dev.topics
	model.eta = dev.eta
	model.mu = dev.mu
	model.sigma = Symmetric(dev.sigma)
	model.invsigma = Symmetric(dev.invsigma)
	model.kappa = dev.kappa
	model.beta = dev.beta
	model.lambda = dev.lambda
	model.lambda_old = deepcopy(model.lambda)
	model.vsq = dev.vsq
	model.logzeta = dev.logzeta
	model.phi = dev_phi1(dev)
	model.tau = dev.tau
	model.tau_old = deepcopy(model.tau)
	model.elbo = dev.elbo
	model.beta ./= sum(model.beta, dims=2); model.beta_old = copy(model.beta)
	model.kappa ./= sum(model.kappa); model.kappa_old = copy(model.kappa)
	nothing
end

# Topic counts the engine takes (tmvb_*_create returns TMVB_EINVAL beyond them, which tmvb_check turns into the ArgumentError the
# reference throws for bad arguments): LDA / fLDA 1024, CTPF 512, CTM / fCTM 256 (round 4; 128 before).  `@gpu train!` on a larger model trains on the CPU
# path with a warning instead of failing -- the macro stays a drop-in for every model the package can build.
hip_max_topics(::Union{LDA, fLDA}) = 1024
hip_max_topics(::Union{CTM, fCTM}) = 256
hip_max_topics(::CTPF) = 512

hipmodel(model::LDA) = hipLDA(model)
hipmodel(model::CTM) = hipCTM(model)
hipmodel(model::CTPF) = hipCTPF(model)
hipmodel(model::fLDA) = hipfLDA(model)
hipmodel(model::fCTM) = hipfCTM(model)

"""
    @gpu train!(model; kwargs...)

Train a topic model on the GPU (MI355X, libtmvb_hip.so).
"""
macro gpu(expr::Expr)
	expr.args[1] == :train! || throw(ArgumentError("GPU acceleration only applies to the train! function."))

	quote
		local model = $(esc(expr.args[2]))
		local kwargs = [(kw.args[1], eval(kw.args[2])) for kw in $(esc(expr.args[3:end]))]

		if isa(model, Union{LDA, CTM, CTPF, fLDA, fCTM}) && model.K > hip_max_topics(model)
			@warn "libtmvb_hip takes K <= $(hip_max_topics(model)) for this model; training on the CPU path."
			train!(model; kwargs...)
		elseif isa(model, Union{LDA, CTM, CTPF, fLDA, fCTM})   # the reference does nothing for fLDA / fCTM (src/macros.jl:274-278)
			local dev = hipmodel(model)
			train!(dev; kwargs...)
			copyback!(model, dev)
			finalize(dev)                        # release the device state now, not at the next GC
			nothing
		else
			train!(model; kwargs...)
		end
	end
end
