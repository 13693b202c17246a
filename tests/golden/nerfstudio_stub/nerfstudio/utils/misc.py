"""nerfstudio.utils.misc: scale_dict (model.py:673)."""


def scale_dict(dictionary, coefficients):
    for key in dictionary:
        if key in coefficients:
            dictionary[key] *= coefficients[key]
    return dictionary
