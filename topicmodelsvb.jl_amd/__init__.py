"""
topicmodelsvb.jl_amd -- MI355X (gfx950) variational-inference engine behind TopicModelsVB.jl's
`TopicModel / train! / @gpu` surface.  Product code: HIP kernels + C ABI in csrc/ (built into
libtmvb_hip.so), and this thin host mirror of the reference's operator interface.

Import through the `tmvb_amd` shim at the repo root (the directory name is not a legal module name).
"""
from ._lib import (CorpusError, DocumentError, EngineError, TopicModelError, build, exported_symbols, lib, LIB_PATH)
from .corpus import (Corpus, Document, PackedCorpus, check_corp, check_doc, dirichlet_rows, readcorp, readcorp_packed, syn_citeu,
                     syn_nsf, synthetic_lda_corpus, writecorp)
from .lda import LDA, DeviceContext, DeviceCorpus, check_model, gpuLDA, gpu_train, predict, topicdist
from .ctm import CTM, check_model_ctm, gpuCTM, gpu_train_ctm, predict_ctm, topicdist_ctm
from .ctpf import CTPF, check_model_ctpf, gpuCTPF, gpu_train_ctpf, topicdist_ctpf
from .comm import Communicator, rccl_version
from .flda import fLDA, check_model_flda, gpufLDA, gpu_train_flda, predict_flda
from .fctm import fCTM, check_model_fctm, gpufCTM, gpu_train_fctm, predict_fctm

__all__ = ["CorpusError", "DocumentError", "EngineError", "TopicModelError", "build", "exported_symbols", "lib", "LIB_PATH",
           "Corpus", "Document", "PackedCorpus", "check_corp", "check_doc", "dirichlet_rows", "readcorp", "readcorp_packed", "writecorp",
           "syn_citeu", "syn_nsf", "synthetic_lda_corpus", "LDA", "DeviceContext", "DeviceCorpus", "check_model", "gpuLDA",
           "gpu_train", "predict", "topicdist", "predict_ctm", "topicdist_ctm", "CTM", "check_model_ctm", "gpuCTM", "gpu_train_ctm", "CTPF", "check_model_ctpf", "gpuCTPF", "gpu_train_ctpf", "Communicator", "rccl_version", "fLDA", "check_model_flda", "gpufLDA", "gpu_train_flda", "fCTM", "check_model_fctm", "gpufCTM", "gpu_train_fctm", "predict_flda", "predict_fctm", "topicdist_ctpf"]
