# gpu_macro.jl -- `macro gpu` on the HIP engine: replaces src/macros.jl:106-284.
#
# The reference's macro builds an empty gpuLDA / gpuCTM / gpuCTPF, assigns every field of the host model to it,
# rebuilds phi (and xi) on the host from the *_old fields, trains, and copies the fields back.  The hip types are
# constructed FROM the host model (they take its fields over, src/macros.jl:115-134 / :153-175 / :196-229) and never
# materialise phi / xi -- the device rebuilds the last-sweep responsibilities from the *_old state inside update_elbo!
# (src/LDA.jl:87-88, src/CTM.jl:93, src/CTPF.jl:239-240) -- so each branch is: construct, train!, copy back.
# The copy-back keeps the reference's post-conditions: *_old = copy of the new value, beta re-normalised in Float64
# (src/macros.jl:147-148, :188-189) so that check_model's isprobvec holds for the Float32-derived rows.
#
# (Julia is not installed in this repository's build image; see TMVBHip.jl.)

"Last-sweep phi of the first document, as the reference keeps in model.phi (src/macros.jl:144, :149)."
function host_phi1(model::LDA)
	model.M == 0 && return Matrix{Float64}[]
	phi = model.beta_old[:, model.corp[1].terms] .* exp.(model.Elogtheta_old[1])
	return [phi ./ sum(phi, dims=1)]
end

function copyback!(model::LDA, dev::hipLDA)
	model.topics, model.alpha, model.beta = dev.topics, dev.alpha, dev.beta
	model.Elogtheta, model.gamma, model.elbo = dev.Elogtheta, dev.gamma, dev.elbo
	model.phi = host_phi1(model)                                   # from the pre-copy *_old state, as :144 does
	model.Elogtheta_old = deepcopy(model.Elogtheta)                # :142
	model.beta ./= sum(model.beta, dims=2)                         # :147
	model.beta_old = copy(model.beta)                              # :148
	nothing
end

function copyback!(model::CTM, dev::hipCTM)
	model.topics, model.mu = dev.topics, dev.mu
	model.sigma, model.invsigma = Symmetric(dev.sigma), Symmetric(dev.invsigma)        # :179-180
	model.beta, model.lambda, model.vsq, model.logzeta, model.elbo = dev.beta, dev.lambda, dev.vsq, dev.logzeta, dev.elbo
	model.lambda_old = deepcopy(model.lambda)                      # :183
	model.beta ./= sum(model.beta, dims=2)                         # :188
	model.beta_old = copy(model.beta)                              # :189
	nothing
end

function copyback!(model::CTPF, dev::hipCTPF)
	model.topics, model.scores, model.drecs, model.urecs = dev.topics, dev.scores, dev.drecs, dev.urecs
	model.alef, model.he, model.bet, model.vav, model.dalet, model.het = dev.alef, dev.he, dev.bet, dev.vav, dev.dalet, dev.het
	model.gimel, model.zayin, model.elbo = dev.gimel, dev.zayin, dev.elbo
	model.alef_old, model.he_old = copy(model.alef), copy(model.he)                    # :247-250
	model.bet_old, model.vav_old, model.dalet_old, model.het_old = copy(model.bet), copy(model.vav), copy(model.dalet), copy(model.het)
	model.gimel_old, model.zayin_old = deepcopy(model.gimel), deepcopy(model.zayin)    # :256-258
	nothing
end

"Last-sweep phi of the first document of a filtered model (src/fLDA.jl:204-207, src/fCTM.jl:230-233)."
function host_phi1(model::fLDA)
	model.M == 0 && return Matrix{Float64}[]
	terms = model.corp[1].terms
	return [additive_logistic(model.tau_old[1]' .* log.(@boink model.beta_old[:,terms]) .+ model.Elogtheta_old[1], dims=1)]
end

function host_phi1(model::fCTM)
	model.M == 0 && return Matrix{Float64}[]
	terms = model.corp[1].terms
	return [additive_logistic(model.tau_old[1]' .* log.(@boink model.beta_old[:,terms]) .+ model.lambda_old[1], dims=1)]
end

function copyback!(model::fLDA, dev::hipfLDA)
	model.topics, model.eta, model.alpha, model.kappa, model.beta = dev.topics, dev.eta, dev.alpha, dev.kappa, dev.beta
	model.Elogtheta, model.gamma, model.tau, model.elbo = dev.Elogtheta, dev.gamma, dev.tau, dev.elbo
	model.phi = host_phi1(model)                                   # from the pre-copy *_old state
	model.Elogtheta_old, model.tau_old = deepcopy(model.Elogtheta), deepcopy(model.tau)
	model.beta ./= sum(model.beta, dims=2); model.beta_old = copy(model.beta)          # Float64 re-normalisation as :147-148
	model.kappa ./= sum(model.kappa); model.kappa_old = copy(model.kappa)
	nothing
end

function copyback!(model::fCTM, dev::hipfCTM)
	model.topics, model.eta, model.mu = dev.topics, dev.eta, dev.mu
	model.sigma, model.invsigma = Symmetric(dev.sigma), Symmetric(dev.invsigma)
	model.kappa, model.beta, model.lambda, model.vsq, model.logzeta = dev.kappa, dev.beta, dev.lambda, dev.vsq, dev.logzeta
	model.tau, model.elbo = dev.tau, dev.elbo
	model.phi = host_phi1(model)
	model.lambda_old, model.tau_old = deepcopy(model.lambda), deepcopy(model.tau)
	model.beta ./= sum(model.beta, dims=2); model.beta_old = copy(model.beta)
	model.kappa ./= sum(model.kappa); model.kappa_old = copy(model.kappa)
	nothing
end

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

		if isa(model, Union{LDA, CTM, CTPF, fLDA, fCTM})       # the reference does nothing for fLDA / fCTM (src/macros.jl:274-278)
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
