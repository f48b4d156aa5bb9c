"""Submodule invocator, like /root/reference/clairvoyante.py (:12-45):

    python -m clairvoyante_amd SubmoduleName [options of the submodule]
"""
import importlib
import sys

SUBMODULES = ("callVarBamParallel", "callVarBam", "callVar", "calTrainDevDiff", "evaluateListOfModels", "evaluate",
              "tensor2Bin", "trainNonstop", "train", "trainWithoutValidationNonstop", "CreateTensor",
              "ExtractVariantCandidates", "GetTruth")
NOT_BUILT = ("demoRun", "getEmbedding", "getTensorAndLayerPNG", "ChooseItemInBed", "CombineMultipleDatasetsForTraining",
             "CountNumInBed", "PairWithNonVariants", "RandomSampling")


def main():
    if len(sys.argv) <= 1:
        print("Clairvoyante (MI355X) submodule invocator:")
        print("  Usage: python -m clairvoyante_amd SubmoduleName [Options of the submodule]")
        print("")
        print("Available submodules:")
        for n in SUBMODULES:
            print("  - %s" % n)
        print("")
        print("Reference submodules outside this build: %s" % ", ".join(NOT_BUILT))
        sys.exit(0)
    name = sys.argv[1]
    if name not in SUBMODULES:
        sys.exit("unknown submodule %r%s" % (name, " (not part of this build)" if name in NOT_BUILT else ""))
    mod = importlib.import_module("clairvoyante_amd.%s" % name)
    sys.argv = sys.argv[1:]
    sys.argv[0] += ".py"
    mod.main()


if __name__ == "__main__":
    main()
