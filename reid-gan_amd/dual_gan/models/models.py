"""create_model(opt) — mirror of CC/dual_gan/models/models.py:4-22."""
import dual_gan.models as models


def create_model(opt):
    model = models.find_model_using_name(opt.model)(opt)
    if opt.verbose:
        print("model [%s] was created" % (model.name()))
    return model
